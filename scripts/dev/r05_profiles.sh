#!/bin/bash
# round 5: everything the committed profiles/r05_* files come from, in one gpurun call
tag=${1:-r05p}
scripts/profile_round.sh $tag
scripts/profile_pmc.sh $tag
