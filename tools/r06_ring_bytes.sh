#!/bin/bash
# The ring's byte budget (include/kpn_dev.hpp, default_ring_bytes_ref) against no budget, same binary, same box, alternating:
# every hot block through the operator API at message sizes either side of the last-level cache.   bash tools/r06_ring_bytes.sh [tag]
tag=${1:-r06}
out=gpurun_out/${tag}_kpn_ring_bytes.txt
: > $out
B="fft:13:20000 fft:16:20000 fft:18:20000 fft:20:10000 fft:21:8000 fft:22:6000 fft:23:4000 fft:24:3000 fft:25:1500 fft:26:800 fft:28:300 channelizer:20:10000 channelizer:22:6000 channelizer:24:3000 channelizer:26:800 channelizer:28:200 fir:22:6000 fir:24:3000 fir:26:800 fir:28:300 ovsave:22:3000 ovsave:24:1000 ovsave:26:300 ovsave:28:100"
C="13:20000:4:0:0:resident:checksum 16:20000:4:0:0:resident:checksum 20:20000:4:0:0:resident:checksum 22:8000:4:0:0:resident:checksum 24:4000:4:0:0:resident:checksum 26:1000:4:0:0:resident:checksum 28:300:4:0:0:resident:checksum 16:20000:4:0:0:carried:checksum 24:4000:4:0:0:carried:checksum 24:4000:4:0:0:synth:checksum"
for rep in 1 2; do
  for mib in 128 4194304; do
    echo "budget_mib $mib rep $rep" >> $out
    KPN_DEV_RING_MIB=$mib tests/_build/kpn_tests bench_block_list $B >> $out 2>&1
    KPN_DEV_RING_MIB=$mib tests/_build/kpn_tests bench_c2_list $C >> $out 2>&1
  done
done
