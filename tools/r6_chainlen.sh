#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export PYTHONUNBUFFERED=1
timeout 300 python tools/gpu_k8_chain_lengths.py 1920 1080 10 3
timeout 300 python tools/gpu_k8_chain_lengths.py 1242 375 24 2
