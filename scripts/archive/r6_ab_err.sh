cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for lib in rows17 dirs17; do
  NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_$lib.so timeout 900 python scripts/onestep_error.py --samples 64 2>&1 | grep '^{' | cut -c1-600
  NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_$lib.so timeout 900 python scripts/onestep_error.py --samples 64 --terrain mixed 2>&1 | grep '^{' | cut -c1-600
done
