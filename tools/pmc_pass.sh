#!/bin/bash
# usage: tools/pmc_pass.sh <outdir> <counters...> -- <bench args...>   (one rocprofv3 --pmc pass; counters in their own run)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d "$GRAFT_REPO_ROOT/$out" -o pmc -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" > "$GRAFT_REPO_ROOT/$out.log" 2>&1
