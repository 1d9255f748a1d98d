#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc.sh <outdir> "<counter list>" -- python3 tools/conv_micro.py ...
# one rocprofv3 --pmc pass (counters only, no tracing), CSV output
out=$1; shift; ctr=$1; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/$out -- "$@" > $R/gpurun_out/$out.log 2>&1
