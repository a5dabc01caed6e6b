(cd _old && python tools/step1_time.py 2>/dev/null && python tools/step1_time.py --task 2>/dev/null)
python tools/step1_time.py 2>/dev/null && python tools/step1_time.py --task 2>/dev/null
echo stage2; RIAB_HIP_LIB=ratinabox_amd/lib/variants/libs1_stage2.so python tools/step1_time.py 2>/dev/null && RIAB_HIP_LIB=ratinabox_amd/lib/variants/libs1_stage2.so python tools/step1_time.py --task 2>/dev/null
(cd _old && python tools/step1_time.py 2>/dev/null && python tools/step1_time.py --task 2>/dev/null)
python tools/step1_time.py 2>/dev/null && python tools/step1_time.py --task 2>/dev/null
echo stage2; RIAB_HIP_LIB=ratinabox_amd/lib/variants/libs1_stage2.so python tools/step1_time.py 2>/dev/null && RIAB_HIP_LIB=ratinabox_amd/lib/variants/libs1_stage2.so python tools/step1_time.py --task 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_step1.py tests/test_gpu_step1_device.py -x -q 2>&1 | tail -15
