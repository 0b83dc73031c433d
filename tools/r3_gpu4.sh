#!/bin/bash
for v in dma_45 dma_55 dma_35 dma_44 dma_45 mx5_222246 mx5_111366 mx_111121 mx_001123 dma_45; do
  SEG_LIB=tools/dbg/libdisn_$v.so timeout -k 10 120 python -u tools/time_lib.py 2>&1 | grep "res 256" || exit 1
done
