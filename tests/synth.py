"""The synthetic input generator lives in the package (microflow_rs_amd/synth.py) so that bench.py and smoke() do not
depend on tests/; the tests keep importing it from here."""
from microflow_rs_amd.synth import SEED, layer_checksum, splitmix64, structured_images, synth_i8  # noqa: F401
