#!/bin/bash
# round 3, call A: the new tests first, then the whole GPU suite file by file, then the default bench (no CPU baseline)
mkdir -p gpurun_out
: > gpurun_out/r03_tests_a.log
for f in tests/test_gpu_checkpoint.py::test_fixture_not_written_by_this_repo_through_from_pretrained tests/test_gpu_caption_batch.py::test_overlapped_stream_with_a_lazy_clip_source_on_the_current_stream \
         tests/test_gpu_kernels.py::test_tome_match_hand_over_stress_alone_and_beside_a_decode_stream tests/test_gpu_configs.py::test_vit_h_free_running_index_audit \
         tests/test_asan_host.py tests/test_gpu_llm.py::test_projector_splice_and_whole_path tests/test_gpu_bench_launch.py::test_config_presets_and_self_diagnosing_fields; do
  echo "=== $f" >> gpurun_out/r03_tests_a.log
  timeout 1500 python -m pytest "$f" -q -m gpu -x -s --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -25 >> gpurun_out/r03_tests_a.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r03_tests_a.log
echo "=== full suite" >> gpurun_out/r03_tests_a.log
for f in tests/test_gpu_*.py; do
  echo "--- $f" >> gpurun_out/r03_tests_a.log
  timeout 1500 python -m pytest $f -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/r03_tests_a.log
done
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
tail -c 600 gpurun_out/r03_bench_a.json
