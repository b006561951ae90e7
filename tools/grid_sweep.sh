for B in 768 1024 1536 2048; do for G in 512 768 1024 1536; do
echo -n "B=$B G=$G: "; EP_POOL_GRID=$G python tools/pool_microbench.py --bwd --B $B | cut -c1-200; done; done
