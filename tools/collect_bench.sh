#!/bin/bash
# Copy the lines of tools/bench_all.sh <tag> from gpurun_out/ into profiles/ (JSON lines only: the gloo rehearsals print
# connection messages on stdout before theirs).
TAG=${1:?tag}
for W in "" _gff _rotor_hmc _quartic_hmc _ho_hmc _quartic_mlmc _quartic_mlmc_hier _rotor_sweep; do grep '^{' gpurun_out/bench_${TAG}$W.json | tail -1 > profiles/${TAG}_bench$W.json; done
grep '^{' gpurun_out/bench_${TAG}_n2.json | tail -1 > profiles/${TAG}_bench_n2_gloo_rehearsal.json
grep '^{' gpurun_out/bench_${TAG}_mlmc_n2.json | tail -1 > profiles/${TAG}_bench_mlmc_n2_gloo_rehearsal.json
cp gpurun_out/driver_${TAG}_b32.json profiles/${TAG}_driver_b32.json
cp gpurun_out/driver_${TAG}_b1.json profiles/${TAG}_driver_b1.json
[ -f gpurun_out/zscores.json ] && cp gpurun_out/zscores.json profiles/${TAG}_zscores.json
echo collected
