"""Test infrastructure only: CPU restatements of the reference hot path used as the parity
checker.  Nothing under ``efficient_probing_amd`` imports this package."""
