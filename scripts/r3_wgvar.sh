#!/bin/bash
set -u
V=metadata-augmented-unet-for-lst-ndvi_amd/variants
ROUNDS=3 timeout -k 10 500 python scripts/variant_bench.py wgrad $V/libmau_base.so $V/libmau_hack16.so
