cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r04_gputest_pass$i.txt; cat gpurun_out/r04_gputest_pass$i.txt
done
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
