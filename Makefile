# Convenience targets; `python -c "import __graft_entry__ as g; g.build()"` does the same as `make all`.
all:
	$(MAKE) -C legion-1_amd/csrc -j8 liblegion_amd.so
	$(MAKE) -C legion-1_amd/csrc legion
	$(MAKE) -C oracle
	cd legion-1_amd/ipc_service && python setup.py build_ext --inplace

test-cpu:
	python -m pytest tests -x -q -m "not gpu"

test-gpu:
	python -m pytest tests -x -q -m gpu

bench:
	python bench.py

clean:
	$(MAKE) -C legion-1_amd/csrc clean
	$(MAKE) -C oracle clean
.PHONY: all test-cpu test-gpu bench clean
