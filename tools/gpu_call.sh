#!/bin/bash
# Wrapper around gpurun: stamps the snapshot with the git head it is taken from (.build_head: .git does not travel to the GPU box), so that
# profiles collected there can say which tree they belong to.  usage: tools/gpu_call.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
h=$(git rev-parse --short=12 HEAD)
if ! git diff --quiet HEAD -- . ':!gpurun_out' 2>/dev/null; then h="$h+uncommitted"; fi
echo "$h" > .build_head
exec /usr/local/graft/bin/gpurun "$@"
