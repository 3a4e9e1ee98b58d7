#!/bin/bash
# Round 5's failing factored-head bring-up form (d loss / d g added to the head's tiles by the vector ALU: -DNEFES_FH_VARIANT_VALU_UPDATE)
# with and without the result fences of round 6, through the per-channel backward test.  Build the side libraries first:
#   make -C nefes_amd/csrc -j7 BUILD=build_v1 OUT=../side/libnefes_fhvalu_nofence.so EXTRA="-DNEFES_FH_VARIANT_VALU_UPDATE -DNEFES_NO_RESULTS_FENCE"
#   make -C nefes_amd/csrc -j7 BUILD=build_v2 OUT=../side/libnefes_fhvalu.so EXTRA="-DNEFES_FH_VARIANT_VALU_UPDATE"
#   make -C nefes_amd/csrc -j7 BUILD=build_nf OUT=../side/libnefes_nofence.so EXTRA=-DNEFES_NO_RESULTS_FENCE
for lib in libnefes_fhvalu_nofence.so libnefes_fhvalu.so libnefes_nofence.so; do
  echo "==== $lib"
  NEFES_HIP_LIB=$PWD/nefes_amd/side/$lib NEFES_PARITY_LOG=/dev/null timeout 300 python -m pytest "tests/test_gpu_hazards.py::test_backward_one_upstream_channel_at_a_time" tests/test_gpu_hazards.py::test_train_mode_backward_with_beta_in_the_loss -m gpu -q -s 2>&1 | grep -v "^$" | grep "hazards\|assert\|Error\|passed\|failed" | cut -c1-300
done
