#!/bin/bash
# From the development container: measure the committed HEAD on a GPU box and write profiles/<tag>_*.
# Usage: bash tools/run_profile_round.sh r02      (the working tree must be clean: the box gets a snapshot of it)
set -e
tag=${1:-r02}
cd "$(dirname "$0")/.."
test -z "$(git status --porcelain)" || { echo "commit first: the profiles name the commit they measured"; exit 1; }
head=$(git rev-parse HEAD)
/usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/profile_round.sh $tag > gpurun_out/profile_round.log 2>&1; tail -5 gpurun_out/profile_round.log"
echo "$head" > "gpurun_out/$tag/COMMIT"
python tools/make_profiles.py "$tag"
