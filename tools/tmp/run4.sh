cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -x -q -k sppf 2>&1 | tail -5
rm -f yolo-lite_amd/csrc/sppf_resident.o yolo-lite_amd/csrc/engine.o
make -C yolo-lite_amd/csrc EXTRA=-DYL_STAMPS > /dev/null 2>&1
for d in 0 1 2 3 4 7; do echo "== dbg $d"; YL_SPPF_DBG=$d timeout -k 10 120 python tools/res_stamps.py sppf 32 20 20 5 1 2>/dev/null | grep -v "^ 0\|workgroups"; done
