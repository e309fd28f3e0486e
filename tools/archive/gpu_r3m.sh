mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_elasticity.py -m gpu -x -q > gpurun_out/r3/m_tests.log 2>&1; echo "tests exit $?"
tail -3 gpurun_out/r3/m_tests.log
timeout -k 10 300 python tools/r3_el_iter.py > gpurun_out/r3/el_iter2.log 2>&1; echo "el iter exit $?"
grep variant gpurun_out/r3/el_iter2.log
timeout -k 10 100 python tools/el_prof.py
