#!/bin/bash
# One command: the TUNING build of libowlhip.so = the shipped kernels (-DOWL_TUNING: process-global switches, traces, ablations)
# + the whole-file experiments of tools/experiments/csrc/.  It REPLACES owl-vit-object-detection_amd/libowlhip.so; rebuild the product
# with `python __graft_entry__.py` afterwards.  Run the tools and tests of this directory with OWL_TUNING=1.
set -e
cd "$(dirname "$0")/../.."
OWL_TUNING=1 bash owl-vit-object-detection_amd/csrc/build.sh
echo "tuning build in place; tests: OWL_TUNING=1 python -m pytest tools/experiments/tests -q -m gpu"
