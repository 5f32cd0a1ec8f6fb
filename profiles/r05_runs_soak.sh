# round 5: the server binary at the papers100M shape over THREE epochs, every 97th served batch (+ first / middle / last / validation / test) against the
# OpenMP oracle word for word, batches 0 and 694 of every epoch also against the serial oracle's committed digests, the rows of every fourth soak batch
# against the generator's closed form.   bash profiles/r05_runs_soak.sh
LEGION_TEST_EPOCHS=3 LEGION_TEST_SERVED_EVERY=97 timeout -k 10 1000 python -m pytest tests/test_gpu_full_shape.py -q -k synth_server --durations=1 2>&1 | tail -n 6
