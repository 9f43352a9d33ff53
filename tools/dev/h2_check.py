import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from phnrec_amd import capi, modelgen
from oracle import binding as ob
from tests.util import model_dir
capi.load()
for system in ["PHN_CZ_SPDAT_LCRC_N1500", "PHN_EN_TIMIT_LCRC_N500", "PHN_HU_SPDAT_LCRC_N1500", "PHN_RU_SPDAT_LCRC_N1500"]:
    spec = modelgen.SYSTEMS[system]; nb = spec["nbanks"]
    d = model_dir(system)
    ctx = capi.Lcrc(d, nb)
    o = ob.Oracle(d, nb)
    for n in [1, 7, 16, 33, 100, 700]:
        mel = modelgen.synth_mel(n, nb, seed=n)
        ref = o.posteriors(mel, threads=8)
        ctx.set_arithmetic(0); a = ctx.posteriors(mel)
        ctx.set_arithmetic(1); b = ctx.posteriors(mel)
        print("%s n=%4d  f32 vs oracle %.2e   split-f16 vs oracle %.2e   split vs f32 %.2e" % (system[4:6], n, np.abs(a-ref).max(), np.abs(b-ref).max(), np.abs(a-b).max()), flush=True)
    for n in [4096, 8192, 32768]:
        mel = torch.from_numpy(modelgen.synth_mel(n, nb, seed=1)).cuda()
        post = torch.empty((n, ctx.n_out), device="cuda")
        s = torch.cuda.current_stream()
        ctx.set_timing(False)
        res = []
        for ar in (0, 1):
            ctx.set_arithmetic(ar)
            for _ in range(200): ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(200): ctx.posteriors_device(mel.data_ptr(), n, post.data_ptr(), stream=s.cuda_stream)
            e1.record(s); s.synchronize()
            res.append(e0.elapsed_time(e1) / 200)
        print("%s %6d frames: f32 %.4f ms   split-f16 %.4f ms  (x%.2f)" % (system[4:6], n, res[0], res[1], res[0]/res[1]), flush=True)
    ctx.close()
