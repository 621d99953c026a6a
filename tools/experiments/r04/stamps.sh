python tools/stamps.py 2>&1 | grep -v amdgpu.ids
echo NOQP; CEDR_STAMPS_LIB=stamps_noqp.so python tools/stamps.py 2>&1 | grep "down wave"
