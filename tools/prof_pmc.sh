#!/bin/bash
# usage: tools/prof_pmc.sh <outdir> <counters...> -- <cmd...>   (separate --pmc pass; kernel-trace only)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d "$out" --output-format csv -- "$@"
