#!/bin/bash
# round 6, call i: the half-size lean workgroup (SRH_LEAN_HALF=1: 256 threads, <= 80 KB of LDS, two rollouts per CU) against the product layout
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; cd $GRAFT_REPO_ROOT
echo "== full-size workgroup (SRH_LEAN_HALF=0)"; SRH_LEAN_HALF=0 timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
echo "== half"; SRH_LEAN_HALF=1 timeout 600 python tools/lean_ab.py c2 --check 2>&1 | grep -v "^library" | tail -7
echo "== half, serial wave by slot"; SRH_LEAN_HALF=1 SRH_LEAN_SERIAL_WAVE=1 timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
echo "== half again"; SRH_LEAN_HALF=1 timeout 600 python tools/lean_ab.py c2 2>&1 | grep -v "^library" | tail -3
