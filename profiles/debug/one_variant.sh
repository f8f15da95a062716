#!/bin/bash
# one test under one variant, with the failure text: bash profiles/debug/one_variant.sh "ENV=1" <pytest args>
v=$1; shift
env $v timeout 600 python -m pytest "$@" -q -m gpu -x 2>&1 | grep -v "^$" | tail -${LINES_OUT:-40}
