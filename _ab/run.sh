for v in 2 4; do echo "== MH $v"; A2S_LIB=/root/repo/_ab/liba2s_mh$v.so timeout 300 python tools/conv_split_check.py 64 2>&1 | grep -v amdgpu.ids | tail -11; done
