"""k_snappy_frames on a collated RAD through Google's snappy (pyarrow), 64 KiB chunks: kernel time from AFQ_HOST_TIMING=1."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AFQ_HOST_TIMING"] = "1"
import numpy as np
import pyarrow as pa
pkg = importlib.import_module("alevin-fry_amd"); af = importlib.import_module("alevin-fry_amd.afquant"); sn = importlib.import_module("alevin-fry_amd.synth_native")
r = sn.generate(seed=3, n_cells=int(sys.argv[1]) if len(sys.argv) > 1 else 400, median_reads=30000, sigma=0.6, num_genes=36601, ref_count=199138)
data = r.data.tobytes(); codec = pa.Codec("snappy"); t0 = time.time()
out = bytearray(b"\xff\x06\x00\x00sNaPpY")
for i in range(0, len(data), 65536):
    body = b"\0\0\0\0" + codec.compress(data[i:i + 65536], asbytes=True)
    out += b"\x00" + len(body).to_bytes(3, "little") + body
print(f"input {len(data) / 1e6:.1f} MB, compressed {len(out) / 1e6:.1f} MB (ratio {len(data) / len(out):.2f}), host compress {time.time() - t0:.1f} s", flush=True)
for _ in range(3):
    got = af.snappy_decode_device(bytes(out))
assert got.tobytes() == data
print("decoded bytes equal the input")
