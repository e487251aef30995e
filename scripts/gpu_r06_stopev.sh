#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
set -e
for mode in 0 1 2 3; do
  for rep in 1 2; do TEZIP_EPART=1 TEZIP_EPART_STOPEV=$mode python scripts/dwp_time.py 2>/dev/null | tail -1 | sed "s/^/STOPEV=$mode forced split /"; done
done
TEZIP_EPART=0 python scripts/dwp_time.py 2>/dev/null | tail -1
