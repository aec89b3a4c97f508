# final evidence run of round 3 (C2): kernel stats + PMC + correct + the plain default bench, error rates, end to end CLI
bash tools/run_profiles.sh all | tail -20
bash tools/err_rates.sh > gpurun_out/err/log.txt 2>&1; cat gpurun_out/err/rates.txt
mkdir -p gpurun_out/e2e
timeout -k 10 400 python tools/e2e_cli.py 1000000 2 > gpurun_out/e2e/e2e_1m.txt 2>&1; tail -6 gpurun_out/e2e/e2e_1m.txt | cut -c1-400
E2E_CPU=0 timeout -k 10 500 python tools/e2e_cli.py 10000000 1 > gpurun_out/e2e/e2e_10m.txt 2>&1; tail -4 gpurun_out/e2e/e2e_10m.txt | cut -c1-400
