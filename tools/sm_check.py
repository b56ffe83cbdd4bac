"""Dev tool: one long DEFLATE stream through the chunked path (AHIP_DEBUG=1 shows its decisions)."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import archive_amd
from archive_amd import _native as N
from tools import corpus
N.lib().ahip_init(0)
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = corpus.WIKI if (len(sys.argv) > 2 and sys.argv[2] == "wiki") else corpus.LOG
if len(sys.argv) > 2 and sys.argv[2] == "noise":  # incompressible: zlib stores it, block after block
    import random
    data = random.Random(9).randbytes(mb << 20)
else:
    data = bytes(corpus.text(kind, 8, 0, mb << 20))
c = zlib.compressobj(6, zlib.DEFLATED, -15)
raw = c.compress(data) + c.flush()
print("raw deflate: %d -> %d bytes" % (len(data), len(raw)))
for it in range(2):
    t = time.perf_counter(); z = archive_amd.Inflate(raw); out = z.get_bytes(); dt = time.perf_counter() - t
    print("Inflate: status %d, %.1f ms, %.3f GB/s out (host buffers), ok=%s" % (z.status, dt * 1e3, len(out) / dt / 1e9, out == data))
zs = zlib.compress(data, 6)
t = time.perf_counter(); out = archive_amd.ZLibDecoder().decode_bytes(zs, verify=True); dt = time.perf_counter() - t
print("ZLibDecoder(verify): %.1f ms ok=%s" % (dt * 1e3, out == data))
import gzip
gz = gzip.compress(data, 6)
for it in range(2):
    t = time.perf_counter(); out = archive_amd.GZipDecoder().decode_bytes(gz); dt = time.perf_counter() - t
    print("GZipDecoder, one member: %.1f ms, %.3f GB/s out (host buffers) ok=%s" % (dt * 1e3, len(out) / dt / 1e9, out == data))
two = gz + gzip.compress(data[:5000000], 6) + gzip.compress(b"tail")
out = archive_amd.GZipDecoder().decode_bytes(two)
print("GZipDecoder, long + long + short members: ok=%s" % (out == data + data[:5000000] + b"tail"))
# device-resident timing
import ctypes, torch
d_in = torch.frombuffer(bytearray(gz), dtype=torch.uint8).cuda(); d_out = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda"); olen = ctypes.c_size_t()
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = N.lib().ahip_gzip_decode_device(d_in.data_ptr(), d_in.numel(), d_out.data_ptr(), d_out.numel(), ctypes.byref(olen), None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("gzip_decode_device: rc %d %.1f ms  %.2f GB/s out  (chunks: %d)" % (rc, dt * 1e3, olen.value / dt / 1e9, N.lib().ahip_debug_last_chunks()))
print("device bytes ok:", bytes(d_out[:olen.value].cpu().numpy()) == data)
os.environ["AHIP_NO_SM"] = "1"
t = time.perf_counter(); out = archive_amd.Inflate(raw).get_bytes(); dt = time.perf_counter() - t
print("one-wave path: %.1f ms, %.3f GB/s ok=%s" % (dt * 1e3, len(out) / dt / 1e9, out == data))
