cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for lib in rows17 dirs17; do
  echo "== $lib"
  NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_$lib.so timeout 600 python -m pytest tests/test_hip_parity_r2.py::test_worlds_of_the_full_batch_follow_the_oracle tests/test_hip_parity_r3.py::test_collapsing_flies_with_every_segment_in_contact_step_like_the_oracle "tests/test_hip_parity_r3.py::test_full_size_batches_step_like_the_oracle_from_their_own_states" -m gpu -q -s 2>&1 | grep "PARITY-LEDGER\|passed\|failed" | cut -c1-900
done
