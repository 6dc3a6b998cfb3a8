#!/bin/bash
# round 5, GPU call 1: the new tests (fp16 flavour, multi-pass row orders, packed accumulation) + in-step A/B of the MSDA forms
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp16_flavour_gpu.py tests/test_msda_bordered_gpu.py tests/test_abi_cpu.py -x -q -m "gpu or not gpu" > $O/c1_tests_a.log 2>&1
echo "tests_a rc=$?" | tee -a $O/c1_tests_a.log
timeout 900 python -m pytest "tests/test_transformer_gpu.py::test_fp16_request_runs_config5_shape" "tests/test_decoder_gpu.py::test_fp16_mode_against_fp16_operand_arithmetic" "tests/test_hotpath_gpu.py::test_stress_pyramid_timed_mode_takes_the_level3_resident_kernel" tests/test_encoder_timed_mode_gpu.py -x -q -s > $O/c1_tests_b.log 2>&1
echo "tests_b rc=$?" | tee -a $O/c1_tests_b.log
for pk in 0 2; do
  SDETR_MSDA_PK=$pk timeout 600 python -m pytest "tests/test_encoder_timed_mode_gpu.py::test_timed_mode_is_no_farther_from_fp32_than_the_reference_autocast" -x -q -s > $O/c1_acc_pk$pk.log 2>&1
  echo "acc pk=$pk rc=$?" | tee -a $O/c1_acc_pk$pk.log
done
bash benchmarks/instep_kernel_us.sh "SDETR_MSDA_PK=0" "SDETR_MSDA_PK=1" "SDETR_MSDA_PK=2" > $O/c1_instep.jsonl 2> $O/c1_instep.err
cat $O/c1_instep.jsonl
grep -h "rc=" $O/c1_*.log
