#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
for rep in 1 2 3; do
  bash tools/abl.sh "$@" 2>&1 | grep rows
done
