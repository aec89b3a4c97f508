mkdir -p gpurun_out/r3x && O=gpurun_out/r3x
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_profiles.sh all | tail -18
