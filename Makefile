# Convenience targets; the pieces have their own Makefiles (software-raytracer_amd/csrc, .../host, oracle).
PY ?= python3

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: build
	$(PY) -m pytest tests -x -q -m gpu

bench: build
	$(PY) bench.py

profile:          # on an MI355X: regenerates profiles/r02, profiles/counters.json and profiles/mesh_counts.json
	$(PY) tools/profile_round.py r02

clean:
	$(MAKE) -C software-raytracer_amd/csrc clean
	$(MAKE) -C software-raytracer_amd/host clean
	$(MAKE) -C oracle clean

.PHONY: build test-cpu test-gpu bench profile clean
