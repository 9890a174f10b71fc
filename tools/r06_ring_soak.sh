#!/bin/bash
# Soak of the device-resident graph's rings on the GPU (kpn_tests devring: synth source -> fused chain -> checksum sink, one thread per block, no host
# synchronisation between blocks: a missing dependency or a buffer recycled too early shows as a different checksum).  Every configuration of one message
# size must print the same checksum: ring depth 1 / 2 / 4, shared and per-block streams, byte budget none / 128 MiB / 1 MiB / 0 (one message out),
# patience 50 ms / 1 ms (the budget then gives way all the time).   bash tools/r06_ring_soak.sh > gpurun_out/r06_ring_soak.txt
E=$GRAFT_REPO_ROOT/tests/_build/kpn_tests
for spec in "10403 60000" "1048576 6000" "20971520 600"; do
  set -- $spec
  echo "== messages of $1 samples x $2"
  for policy in 0 1; do for depth in 1 2 4; do for mib in 4194304 128 1 0; do for pat in 50 1; do
    printf "policy %d depth %d budget_mib %-8d patience_ms %-3d " $policy $depth $mib $pat
    KPN_DEV_RING_MIB=$mib KPN_DEV_RING_PATIENCE_MS=$pat timeout 600 $E devring $1 $2 $depth 20 $policy 0 || echo "FAILED rc=$?"
  done; done; done; done
done
