#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 900 python3 -m pytest tests/test_attention_gpu.py -m gpu -q -x -k "page_sizes or leftpad or batch_idx" 2>&1 | tail -6
