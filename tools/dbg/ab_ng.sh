#!/bin/bash
# the dense product's XCD mapping: NG row groups x 8 / NG column-panel classes (R fetched by NG XCDs, 8 / NG x 0.5 MB... of P per XCD)
for v in 4 2 8 1; do echo "== FBX_NG=$v"; FBX_NG=$v python tools/fb_lane_curve.py --dense 20 48 50 64 2>&1 | tail -4 | sed 's/|.*//'; done
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in 4 2; do
  rm -rf /tmp/p_$v
  FBX_NG=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_$v -o run -- python $REPO/tools/fb_lane_curve.py --dense 50 > /tmp/p.log 2>&1
  echo "== FETCH_SIZE FBX_NG=$v"; python $REPO/tools/rocpd_pmc.py $(find /tmp/p_$v -name '*.db' | head -1) | grep -E "k_fb_prodILi1ELi2" | cut -c1-160
done
