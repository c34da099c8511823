"""GPU: image pre / post-processing.  The resize is integer work: byte-exact against the installed Pillow
(the library torchvision's Resize calls for PIL inputs) and against the oracle restatement of its algorithm; the float steps
(ToTensor / Normalize, uint8 conversion) are the same IEEE operations as the reference's torch / numpy calls: bit-exact."""
import numpy as np
import pytest
import torch

import util
from util import O

pytestmark = pytest.mark.gpu


def _img(rs, H, W, C):
    img = rs.randint(0, 256, size=(H, W, C)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    img[..., 0] = (127 + 120 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8)  # smooth content next to noise
    return img


@pytest.mark.parametrize("H,W,C,R", [(300, 451, 3, 256), (1000, 750, 3, 256), (64, 64, 3, 256), (512, 512, 3, 256),
                                     (97, 400, 3, 64), (700, 511, 1, 512), (256, 256, 3, 256)])
def test_image_transform_byte_exact_with_pil_and_oracle(H, W, C, R):
    from PIL import Image
    P = util.pkg()
    rs = np.random.RandomState(H + W)
    img = _img(rs, H, W, C)
    pil = Image.fromarray(img if C == 3 else img[..., 0], "RGB" if C == 3 else "L")
    out, u8 = P.image_utils.image_transform(pil, resolution=R, normalize=True, return_bytes=True)
    want, want_u8 = O.image_transform_np(img, R, True)
    assert tuple(out.shape) == (C, R, R) and out.dtype == torch.float32 and out.is_cuda
    assert np.array_equal(u8.cpu().numpy(), want_u8)               # resized + cropped bytes == oracle
    assert torch.equal(out.cpu(), want)                            # ToTensor / Normalize: same fp32 ops
    # and the oracle's resize is PIL's (checked here against the installed library, not only in make_golden)
    if W <= H:
        ow, oh = R, int(R * H / W)
    else:
        oh, ow = R, int(R * W / H)
    ref = np.asarray(pil.resize((ow, oh), Image.BICUBIC)).reshape(oh, ow, C)
    ct, cl = int(round((oh - R) / 2.0)), int(round((ow - R) / 2.0))
    assert np.array_equal(u8.cpu().numpy(), ref[ct:ct + R, cl:cl + R])
    # un-normalised form (the inpainting mask path) and tensor input
    out2 = P.image_utils.image_transform(torch.from_numpy(img), resolution=R, normalize=False)
    assert torch.equal(out2.cpu(), O.image_transform_np(img, R, False)[0])
    assert float(out2.min()) >= 0.0 and float(out2.max()) <= 1.0


def test_image_transform_golden_fixture():
    g = util.golden("image_ops.npz")
    out, u8 = util.pkg().image_utils.image_transform(torch.from_numpy(g["img"]), resolution=int(g["resolution"]), return_bytes=True)
    assert np.array_equal(u8.cpu().numpy(), g["pil_bytes"]) and np.array_equal(out.cpu().numpy(), g["tensor"])


def test_images_to_uint8_bit_exact():
    P = util.pkg()
    torch.manual_seed(0)
    x = torch.randn(3, 3, 64, 48) * 0.8
    x[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 0.999999, -0.999999, 0.5])
    got = P.image_utils.images_to_uint8(x.cuda())
    want = O.images_to_uint8_np(x)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (3, 64, 48, 3) and np.array_equal(got.cpu().numpy(), want)


def test_inpainting_token_mask_vs_torch_bicubic():
    P = util.pkg()
    R = 256
    yy, xx = np.mgrid[0:R, 0:R]
    for k, m in enumerate([((xx - 120) ** 2 + (yy - 100) ** 2 < 70 ** 2), (xx > 77) & (xx < 190) & (yy > 60) & (yy < 201)]):
        mask = torch.from_numpy(m.astype(np.float32))[None]
        got, val = P.image_utils.inpainting_token_mask(mask.cuda(), R, batch_size=2, return_values=True)
        want, wval = O.inpainting_token_mask(mask, R)
        assert tuple(got.shape) == (2, 256) and got.dtype == torch.bool
        assert (val.cpu() - wval).abs().max() < 1e-5
        safe = (wval - 0.5).abs() > 1e-4  # away from the threshold the boolean decision must agree
        assert torch.equal(got[0].cpu()[safe], want[safe]) and torch.equal(got[0], got[1])
        assert 0 < int(got[0].sum()) < 256
