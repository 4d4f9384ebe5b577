# Convenience targets; the contract is __graft_entry__.build() / smoke(), bench.py and pytest.
PY ?= python3
.PHONY: build test test-gpu bench clean
build:
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build
	$(PY) -m pytest tests -x -q -m "not gpu"
test-gpu: build
	$(PY) -m pytest tests -x -q -m gpu
bench: build
	$(PY) bench.py
clean:
	$(MAKE) -C kaldi_amd/csrc clean 2>/dev/null || rm -rf kaldi_amd/build kaldi_amd/lib
	$(MAKE) -C oracle clean
