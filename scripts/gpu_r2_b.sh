#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
( python scripts/diag_plane_scatter.py full
  LP_MFMA_DEBUG=8 python scripts/diag_plane_scatter.py full
  LP_MFMA_GENERIC_GRIDS=1 python scripts/diag_plane_scatter.py full
  python scripts/diag_plane_scatter.py search ) > gpurun_out/r2_diag_plane.txt 2>&1
tail -60 gpurun_out/r2_diag_plane.txt
