"""Per-forward time of the wide-net configurations, once per kernel variant (KAMI_WIDE_VARIANT is read once per process:
this script re-runs itself per variant).  0 = the launcher's own choice; 1/2/3 = conv_mfma_kernel with that many
workgroups per CU; 4 = conv4_mfma_kernel (four boards x 128 output channels per workgroup), per layer;
5 = tower128_kernel (128 filters: the whole 3x3 stack in one launch, four boards per workgroup); 6 = tower2b_kernel
(128 / 256 filters: the whole stack, two boards per workgroup, waves own output channels)."""
import sys, os, subprocess, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
CASES = ((119, 128, 10, 1024, "bf16"), (119, 128, 10, 512, "bf16"), (119, 128, 10, 2048, "bf16"),
         (119, 256, 20, 256, "f16"), (119, 256, 20, 512, "f16"), (119, 256, 20, 2048, "f16"))
if len(sys.argv) > 1:
    import numpy as np
    from kami_amd import NN, weights as W, _lib as L
    lib = L.load()
    for F, Cc, R, B, dt in CASES:
        nn = NN(8, 8, F, 4672, filters=Cc, residuals=R, dtype=dt)
        nn.load_weights(W.random_weights(F, Cc, R, seed=1), 1)
        x = np.random.default_rng(0).random((B, 8, 8, F), dtype=np.float32)
        d_in = C.c_void_p(); d_p = C.c_void_p(); d_v = C.c_void_p()
        lib.kh_dev_alloc(nn.handle, x.nbytes, C.byref(d_in)); lib.kh_dev_alloc(nn.handle, B*4672*4, C.byref(d_p)); lib.kh_dev_alloc(nn.handle, B*256*4, C.byref(d_v))
        lib.kh_memcpy_h2d(nn.handle, d_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
        r = []
        for _ in range(5):
            ms = C.c_float(); assert lib.kh_time_infer_device(nn.handle, d_in, B, d_p, d_v, max(20, 60 * 512 // B), C.byref(ms)) == 0, L.last_error(); r.append(ms.value)
        med = float(np.median(r[1:]))
        tf = W.flops_per_eval(F, Cc, R) * B / (med * 1e-3) / 1e12
        print(f"variant {sys.argv[1]}: {R}x{Cc} B={B} {dt}: {med*1e3:8.1f} us  {B / (med * 1e-3):12,.0f} evals/s  {tf:7.1f} TFLOP/s  {tf / 2500:.3f}", flush=True)
        for p in (d_in, d_p, d_v): lib.kh_dev_free(nn.handle, p)
        nn.close()
else:
    for v in (0, 6, 5, 3):
        subprocess.run([sys.executable, __file__, str(v)], env=dict(os.environ, KAMI_WIDE_VARIANT=str(v)), check=False)
