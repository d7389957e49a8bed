#!/bin/bash
# round-3 GPU session AW: k_project_u8_max with two plane buffers in turn (253 registers, two workgroups per CU) vs one (164, three)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3aw; mkdir -p $O
RML_LIB=$PWD/radar-ml_amd/libradarml_hip_u8db.so timeout 600 python -m pytest tests/test_projection_gpu.py -x -q -k "uint8 or u8" 2>&1 | tail -n 2
for rep in 1 2 3; do for lib in db base; do
  L=""; [ $lib = db ] && L=$PWD/radar-ml_amd/libradarml_hip_u8db.so
  echo "== $lib $rep"
  RML_LIB=$L timeout 300 python tools/kbench.py proj --u8 --grid 64x64x128 --frames 16384 2>&1 | grep "codes+stats" | cut -c40-200
  RML_LIB=$L timeout 300 python tools/kbench.py proj --u8 --grid 22x31x176 --frames 32768 2>&1 | grep "codes+stats" | cut -c40-200
done; done
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024 --ingest u8"
for rep in 1 2; do for lib in db base; do
  L=""; [ $lib = db ] && L=$PWD/radar-ml_amd/libradarml_hip_u8db.so
  RML_LIB=$L timeout 900 $B > $O/${lib}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/${lib}_$rep.json u8-$lib
done; done
