# dev: shader clock while the bench loop runs (is the part running at its boost clock during the short layer kernels?)
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -5
python bench.py --no-cpu-baseline --no-extras --steps 4000 --warmup 5 > /tmp/b.log 2>&1 &
BP=$!
sleep 20
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -i "sclk" | head -2; rocm-smi --showpower 2>&1 | grep -i "power" | head -2; sleep 0.5; done
wait $BP
tail -1 /tmp/b.log | cut -c1-300
