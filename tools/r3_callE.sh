mkdir -p gpurun_out/r3f && O=gpurun_out/r3f
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py::test_c5_full_size_wide_index_one_shard_of_eight -x -q -s > $O/c5.log 2>&1; echo "c5 rc=$?"; grep -v "^$" $O/c5.log | tail -6 | cut -c1-900
bash tools/r3_shapes.sh c5 bench kt pmc
