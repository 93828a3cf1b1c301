#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line of the library in the tree (B) against another build of it kept as
# diffusionmodelscustom_amd/csrc/ab_base_lib.so (A) on the SAME box, alternating A B A B, since boxes differ by more than most kernel changes.
#   BENCH_ARGS="--timesteps 400" bash tools/ab_bench.sh
# With AB_ENV set (e.g. AB_ENV="DANRA_CONV_S2=0") A is the library in the tree run under that environment instead.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
LIB=diffusionmodelscustom_amd/libdanra_hip.so
mkdir -p gpurun_out
cp "$LIB" /tmp/new.lib
for round in 1 2; do
    for which in base new; do
        EXTRA=""
        if [ $which = base ]; then
            if [ -n "${AB_ENV:-}" ]; then EXTRA="$AB_ENV"; else cp diffusionmodelscustom_amd/csrc/ab_base_lib.so "$LIB"; fi
        else cp /tmp/new.lib "$LIB"; fi
        env $EXTRA timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-kernel-timing --no-strict-fp32 ${BENCH_ARGS:-} > gpurun_out/ab_$which$round.json 2> gpurun_out/ab_$which$round.err
        python3 -c "import json;d=json.load(open('gpurun_out/ab_$which$round.json'));print('$which', $round, round(d['value'],2), 'img/s', round(d['unet_ms'],4), 'ms')"
    done
done
cp /tmp/new.lib "$LIB"
