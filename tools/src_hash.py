#!/usr/bin/env python3
"""sha256 over the kernel sources a profile was taken on (efficient_probing_amd/csrc/*.hip, *.h and include/*.h, by sorted
relative path and content).  tools/make_profiles.sh stores it in every *_hbm_traffic_pmc.json; bench.py recomputes it and
refuses to quote `roofline.traffic` from a profile of other sources."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(root: str = ROOT) -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "efficient_probing_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "efficient_probing_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash(sys.argv[1] if len(sys.argv) > 1 else ROOT))
