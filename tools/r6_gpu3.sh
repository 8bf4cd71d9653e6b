export EEM_LIB_PATH=$PWD/eemflow_amd/libeemflow_hip_diag.so
for d in 0 14 6; do
echo "== EEM_WG_DBG=$d"
EEM_WG_DBG=$d python tools/wgrad_sweep.py 64 64 3 3 1 64 20,40,80,160 24,48,96 2>&1 | grep -v amdgpu.ids
done
