"""Compile csrc/viterbi_tiera.hip with hiprtc under the options of a .defs file (+ extra options) and print the register / scratch
figures of the code object: what does the SAME source give on this machine?   python tools/hiprtc_probe.py <file.defs> [arch] [extra ...]"""
import ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
defs = [l.strip() for l in open(sys.argv[1]) if l.strip()]
arch = sys.argv[2] if len(sys.argv) > 2 else "gfx950"
extra = sys.argv[3:]
src = open(os.path.join(ROOT, "dnastore_amd", "csrc", "viterbi_tiera.hip")).read().encode()
rtc = ctypes.CDLL("libhiprtc.so")
prog = ctypes.c_void_p()
assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src, b"viterbi_tiera.hip", 0, None, None) == 0
opts = ["--offload-arch=" + arch, "-O3", "-ffp-contract=off", "-std=c++17"] + defs + extra
arr = (ctypes.c_char_p * len(opts))(*[o.encode() for o in opts])
rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
n = ctypes.c_size_t()
if rc != 0:
    rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n)); log = ctypes.create_string_buffer(n.value); rtc.hiprtcGetProgramLog(prog, log)
    print("compile failed:", log.value.decode()[-2000:]); sys.exit(1)
rtc.hiprtcGetCodeSize(prog, ctypes.byref(n))
code = ctypes.create_string_buffer(n.value)
rtc.hiprtcGetCode(prog, code)
with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
    f.write(code.raw)
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], stdout=subprocess.PIPE).stdout.decode()
keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("vgpr_count", "vgpr_spill", "sgpr_spill", "private_segment_fixed", "amdhsa.target"))]
print(arch, " ".join(extra), "| size", n.value, "|", "; ".join(keep))
