#!/bin/bash
set -u
for r in 2 4 8 16; do echo "replicas $r"; RSRL_TILE_REPLICAS=$r python scripts/prof_shared.py tile none; done
