#!/bin/bash
# A/B timing of kernel-library variants on the bench workload: tools/ab.sh out_dir variant...
# (a variant `x` is objective-slam_amd/liboslam_x.so, built with make BUILD=build_x OUT=../liboslam_x.so EXTRA=...)
O=$1; shift
mkdir -p $O
for v in "$@"; do
  OSLAM_PROF=1 OSLAM_LIB=$PWD/objective-slam_amd/liboslam_$v.so timeout -k 10 120 python tools/explore.py 5000 100000 8 0.025 ${MODE:-0} > $O/$v.log 2>&1
  echo "== $v: $(grep -h 'align' $O/$v.log | tail -1 | cut -c1-60) $(grep -h 'ms_vote_kernel' $O/$v.log | tail -1 | sed -e "s/.*'ms_vote_kernel': \([0-9.]*\).*'ms_key_kernel': \([0-9.]*\).*/vote \1 key \2/")"
  grep -h "oslam prof" $O/$v.log | tail -1
done
