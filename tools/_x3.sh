for v in ab hip; do echo "--- $v"; QPWC_HIP_LIB=$PWD/qpwcnet_amd/csrc/libqpwc_$v.so python tools/enc16bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-90; done
