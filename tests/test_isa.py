"""What hipcc emits for the headline kernel, looked at: no GPU needed (hipcc cross-compiles gfx950), seconds."""
import os
import subprocess

def test_trunk_kernel_has_no_packed_fp32_instructions(tmp_path):
    """A packed-fp32 instruction gets no issue slot beside the other wave's MFMAs (profiles/r05_coissue/): trunkw_kernel is built
    with __attribute__((target("no-packed-fp32-ops"))) (UVA_NO_PK_F32, csrc/uva_devutil.hip.h).  Should a compiler stop honouring
    that, the library would still be correct and 2 % slower without anybody noticing: the assembly hipcc emits HERE is looked at."""
    from upscale_video_amd import build
    asm = str(tmp_path / "uva_wino.s")
    cmd = [build.hipcc()] + [f for f in build.FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "uva_wino.hip"), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(asm).read()
    body = text[text.index("trunkw_kernel"):]
    mfma = body.count("v_mfma_f32_16x16x32_f16")
    packed = sum(body.count(op) for op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"))
    assert mfma >= 3 * 192, mfma            # three instantiations, two k-loops of 96 each
    assert packed == 0, packed
