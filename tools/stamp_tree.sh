#!/bin/bash
# Run HERE (the container with .git) right before a gpurun call: records the commit the shipped tree belongs to, for the provenance line
# of the PMC summaries collected on the GPU box (tools/pmc_summary.py).  tools/stamp_tree.sh && gpurun -- '...'
root="$(cd "$(dirname "$0")/.." && pwd)"
sha=$(git -C "$root" rev-parse HEAD)
[ -n "$(git -C "$root" status --porcelain --untracked-files=no)" ] && sha="$sha-dirty"
echo "$sha" > "$root/gpurun_stamp.txt"
echo "$sha"
