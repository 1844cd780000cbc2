#!/bin/bash
export TMPDIR=/tmp
for v in 0 3 4; do LP_MFMA_FWD_VARIANT=$v timeout 300 python scripts/fwd_variants.py 2>&1 | tail -1; done
