#!/bin/bash
mkdir -p gpurun_out/s14
python -m pytest tests -m gpu -x -q > gpurun_out/s14/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s14/pytest.log
python bench.py --no-cpu --no-single > gpurun_out/s14/bench.json 2> gpurun_out/s14/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s14/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s14/smoke.log
tail -3 gpurun_out/s14/pytest.log
