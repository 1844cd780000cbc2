#!/bin/bash
export TMPDIR=/tmp
LIGHTPLANE_AMD_LIB=$PWD/ab/libB.so timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -5
bash scripts/gpu_ab.sh
