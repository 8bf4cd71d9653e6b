python -m pytest tests -m gpu -x -q -k "voxel" 2>&1 | tail -2
for bf in 12288 4608; do for tp in 0 8; do
echo "== BAND_FLOATS=$bf TWOPASS=$tp"
EEM_VOX_BAND_FLOATS=$bf EEM_VOX_TWOPASS=$tp timeout 300 python3 tools/voxel_bench.py 200000 2000000 2>&1 | grep "bins=5"
EEM_VOX_BAND_FLOATS=$bf EEM_VOX_TWOPASS=$tp timeout 300 python3 tools/pipe_probe.py 2>&1 | grep "voxelize x2 only \|full pipeline  "
done; done
