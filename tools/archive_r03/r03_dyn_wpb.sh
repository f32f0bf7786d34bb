#!/bin/bash
# k_dynobs block size sweep (tuning): MGX_DYN_WPB waves per block
for w in 1 2 4; do export MGX_DYN_WPB=$w; echo "wpb=$w"; tools/archive_r03/r03_dyn_kernels.sh $1 1048576 svc2 | grep k_dynobs; done
