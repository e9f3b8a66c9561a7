#!/bin/bash
# usage: tools/pmc.sh <outdir> <counter list> -- <program...>   (one PMC pass, csv output)
out=$1; shift; ctr=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- "$@" > $out.log 2>&1
