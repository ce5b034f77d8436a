#!/bin/bash
# Build everything locally (the built .so files travel with the snapshot), then run a command on the MI355X box.
#   tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun "$@"
