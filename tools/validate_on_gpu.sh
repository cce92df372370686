cd $GRAFT_REPO_ROOT
rm -f gpurun_out/gpu_parity_report.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
echo "all gpu tests rc=$?" >> gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 1500 python bench.py > gpurun_out/b_full2.json 2> gpurun_out/b_full2.err
echo "full rc=$?" >> gpurun_out/b_full2.err
tail -4 gpurun_out/t_all.log; tail -2 gpurun_out/smoke.log; tail -3 gpurun_out/b_full2.err
