# Convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly).
PY ?= python

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: build
	$(PY) -m pytest tests -x -q -m "not gpu"

gpu-test:
	$(PY) -m pytest tests -x -q -m gpu

smoke:
	$(PY) __graft_entry__.py --smoke

bench:
	$(PY) bench.py

profile:
	bash scripts/profile_gpu.sh manual

clean:
	rm -f suchtree_amd/libsuchtree_hip.so oracle/liboracle.so oracle/liboracle_asan.so tests/emu/libst_emu.so

.PHONY: build test gpu-test smoke bench profile clean
