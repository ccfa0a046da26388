#!/bin/bash
set -u
bash scripts/gpu_tests.sh
bash scripts/gpu_prof_shared.sh
