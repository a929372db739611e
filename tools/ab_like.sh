#!/bin/bash
# Runs ON THE GPU BOX: likelihood launch times of two builds of the library side by side (tools/kbench.py --lib).
# usage: bash tools/ab_like.sh <other .so> [kbench arguments]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=$1; shift
echo "== this build"; timeout -k 10 300 python3 $R/tools/kbench.py "$@" || exit 1
echo "== $OTHER"; timeout -k 10 300 python3 $R/tools/kbench.py --lib $R/$OTHER "$@" || exit 1
