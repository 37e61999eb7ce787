#!/bin/bash
# round 6, session 25: leaf-packing models of general leaf widths (hens_rj_set_model_general)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_rj_callable.py -q 2>&1 | tail -25
