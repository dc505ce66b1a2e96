#!/bin/bash
# Usage (GPU box, repo root, library built with `tools/build_exp.sh`, VPPX_LIB=tools/bin/libvppx_exp.so): tools/exp_we.sh <outdir>
# What precedes the W/E launch of a step, one configuration per process (tools/we_probe.py).
out=${1:-gpurun_out/exp_we}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for cfg in "X=0" "VPPX_EXP_PRE=1" "VPPX_EXP_PRE=2" "VPPX_EXP_PRE=3" "VPPX_EXP_PRE=6" "VPPX_EXP_ORDER=1" "VPPX_EXP_TWICE=1"; do
    env $cfg python3 tools/we_probe.py 32 10 >> $out/we_probe.jsonl 2>> $out/we_probe.err
done
cat $out/we_probe.jsonl
