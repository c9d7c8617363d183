mkdir -p gpurun_out/ab
make -s -C build/head_csrc -B OUT=/tmp/librtw_head.so 2>&1 | grep error
cp raytracingweekend.jl_amd/lib/librtw_hip.so /tmp/librtw_tree.so
for r in 1 2 3; do for v in head tree; do echo "$v mfma 1000spp: $(RTW_HIP_LIB=/tmp/librtw_$v.so RTW_DRAIN_PROFILE=1 timeout 300 python tools/gpu_quick.py f32 1920 1000 50 plain 2 2>&1 | grep -E "drain profile\] [0-9]+ waves" | tail -1 | sed 's/.*kernel span/span/')"; done; done > gpurun_out/ab/valu6.txt 2>&1
for v in head tree; do echo "$v valu 300spp: $(RTW_SCAN=valu RTW_HIP_LIB=/tmp/librtw_$v.so RTW_DRAIN_PROFILE=1 timeout 300 python tools/gpu_quick.py f32 1920 300 50 plain 2 2>&1 | grep -E "drain profile\] [0-9]+ waves" | tail -1 | sed 's/.*kernel span/span/')"; done >> gpurun_out/ab/valu6.txt 2>&1
cat gpurun_out/ab/valu6.txt
