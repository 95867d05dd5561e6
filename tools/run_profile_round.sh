#!/bin/bash
# From the development container: measure the committed HEAD on a GPU box and write profiles/<tag>_*.
# Usage: bash tools/run_profile_round.sh r03      (the working tree must be clean: the box gets a snapshot of it)
set -e
tag=${1:-r03}
cd "$(dirname "$0")/.."
test -z "$(git status --porcelain)" || { echo "commit first: the profiles name the commit they measured"; exit 1; }
head=$(git rev-parse HEAD)
make -C omr-img-corrector_amd/csrc all debug > /dev/null   # (tools/kstamps.py and hstamps.py load the debug library)
mkdir -p "gpurun_out/$tag"
for part in a b c; do
  /usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/profile_round.sh $tag $part > gpurun_out/profile_round_$part.log 2>&1; tail -3 gpurun_out/profile_round_$part.log"
done
echo "$head" > "gpurun_out/$tag/COMMIT"
python tools/make_profiles.py "$tag"
