#!/bin/bash
# Build the library at a git revision (default HEAD) next to the working-tree one, for same-box A/B timing:
#   bash tools/exp/ab_build.sh [rev] [suffix]   ->  mppi-isaac_amd/csrc/libmppi_hip_<suffix>.so   (suffix: base)
# then e.g.  python tools/exp/scene_breakdown.py mppi-isaac_amd/csrc/libmppi_hip_base.so
set -e
rev=${1:-HEAD}
suffix=${2:-base}
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
git -C "$root" worktree add -f --detach "$tmp/w" "$rev" >/dev/null 2>&1
(cd "$tmp/w" && python __graft_entry__.py >/dev/null 2>&1)
cp "$tmp/w/mppi-isaac_amd/csrc/libmppi_hip.so" "$root/mppi-isaac_amd/csrc/libmppi_hip_$suffix.so"
git -C "$root" worktree remove --force "$tmp/w"
rm -rf "$tmp"
echo "built $rev -> mppi-isaac_amd/csrc/libmppi_hip_$suffix.so"
