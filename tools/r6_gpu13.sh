echo "== bx3 at MT = 1 (EEM_WGRAD_BX3_MT1=1) against the default"
python tools/wgrad_bench.py 20 "16->16" 2>&1 | grep -v amdgpu.ids
EEM_WGRAD_BX3_MT1=1 python tools/wgrad_bench.py 20 "16->16" 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py 2>&1 | grep -v amdgpu.ids; EEM_WGRAD_BX3_MT1=1 python tools/bench_train.py 2>&1 | grep -v amdgpu.ids
echo "== FETCH_SIZE / WRITE_SIZE per frame at 10, 3 and 1 frames per launch (how much of FETCH_SIZE is Infinity-Cache traffic)"
bash tools/pmc_batch.sh r06_pmc10 10
bash tools/pmc_batch.sh r06_pmc3 3
bash tools/pmc_batch.sh r06_pmc1 1
