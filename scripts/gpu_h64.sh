#!/bin/bash
export TMPDIR=/tmp
HID=64 timeout 600 python scripts/bench_h64.py 2>&1 | tail -1
HID=32 timeout 600 python scripts/bench_h64.py 2>&1 | tail -1
