"""robust_mvd's 2-D CNN on the engine: every convolution of the DispNet blocks (encoder, context encoder, fusion score
convs, cost-volume encoder, decoder; /root/reference rmvd/models/robust_mvd.py:57-99 and the blocks it calls:
dispnet_encoder.py:6-27, dispnet_context_encoder.py, learned_fusion.py:8-48, dispnet_costvolume_encoder.py:7-50,
dispnet_decoder.py:36-138) runs on ops.conv2d_split (split-operand fp16 MFMA, fp32-grade), activations stay channel-last
from the first layer to the prediction heads, the sweep (K1) and the fusion arithmetic (K2) read and write those layouts
directly, and every `torch.cat` of the reference is a layer writing its channel slice of the consumer's input buffer.

Buffers of one forward (n = batch, V source views, h8 = H/8 ...; channel counts padded to a multiple of 8 with zero channels):
  cat5 (n,H/2,W/2,104)  = [conv1(key) 64 | deconv_5 32 | up(pred_4) 2 | 0]      -> rfeat5
  cat4 (n,H/4,W/4,200)  = [conv2(key) 128 | deconv_4 64 | up(pred_3) 2 | 0]     -> rfeat4
  merged (n,h8,w8,288)  = [conv_redir 32 | fused correlation 256]               -> conv3_1
  cat3 (n,h8,w8,392)    = [conv3_1 256 | deconv_3 128 | up(pred_2) 2 | 0]       -> rfeat3
  cat2 (n,h16,w16,776)  = [conv4_1 512 | deconv_2 256 | up(pred_1) 2 | 0]       -> rfeat2
  cat1 (n,h32,w32,1032) = [conv5_1 512 | deconv_1 512 | up(pred_0) 2 | 0]       -> rfeat1
The source views' conv3 output is written into the interior of zero-bordered maps, the layout K1 samples from.
Each buffer has one max-|x| slot (a device float) that all its producers raise atomically; the consuming layer scales its
activations by it (see csrc/conv2d_split.hip)."""
import torch

from . import _lib as L
from . import ops


class DispnetEngine:
    """Packed weights + the forward of RobustMVD's network part on the engine.  Built lazily by RobustMVD, re-packed when a
    parameter changes."""

    def __init__(self, model):
        self.model = model
        self._key = None
        self.w = None
        self._bufs = {}
        self._sides = {}
        self.side_stream = True  # key view's encoder chain beside the source views' (tools/path_a_engine.py measures both)

    # ------------------------------------------------------------------------------------------------------------------
    def _prepare(self):
        m = self.model
        key = tuple((p.data_ptr(), p._version) for p in m.parameters())
        if self.w is not None and self._key == key:
            return self.w
        w = {}

        def pack(name, conv, mode=L.CONV2D, cin_pad=None):
            stride = conv.stride[0]
            w[name] = ops.pack_conv2d_weights_split(conv.weight.detach(), conv.bias, stride=stride, mode=mode, cin_pad=cin_pad)

        enc, fe, dec = m.encoder, m.fusion_enc_block, m.decoder
        pack("conv1", enc.conv1[0], L.CONV2D_IMAGE, 8)
        pack("conv2", enc.conv2[0])
        pack("conv3", enc.conv3[0])
        pack("conv_redir", m.context_encoder.conv_redir[0])
        pack("score3", m.fusion_block.corr_to_view_weight[0])
        pack("score1", m.fusion_block.corr_to_view_weight[2])
        for name in ("conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1"):
            pack(name, getattr(fe, name)[0])
        pack("pred_0", dec.pred_0[0])
        for lvl in range(1, 6):
            pack(f"deconv_{lvl}", getattr(dec, f"deconv_{lvl}")[0], L.DECONV2D)
            conv = getattr(dec, f"rfeat{lvl}")[0]
            pack(f"rfeat{lvl}", conv, cin_pad=(conv.in_channels + 7) // 8 * 8)
            pack(f"pred_{lvl}", getattr(dec, f"pred_{lvl}")[0])
        self.w, self._key = w, key
        return w

    def _buffers(self, n, H, W, V, dev):
        """The buffers whose pad channels / borders must read zero, allocated once per (shape, stream) and reused: every forward
        overwrites everything else in them.  Keyed by the stream too: FramePipeline runs consecutive frames on two streams."""
        key = (n, H, W, V, str(dev), torch.cuda.current_stream(dev).cuda_stream)
        b = self._bufs.get(key)
        if b is None:
            if len(self._bufs) >= 8:  # (shape, stream) combinations come and go: drop them all, but only once nothing in flight uses them
                torch.cuda.synchronize(dev)
                self._bufs.clear()
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
            h8, w8 = H // 8, W // 8
            b = self._bufs[key] = {
                "cat5": z(n, H // 2, W // 2, 104), "cat4": z(n, H // 4, W // 4, 200), "cat3": z(n, h8, w8, 392),
                "cat2": z(n, h8 // 2, w8 // 2, 776), "cat1": z(n, h8 // 4, w8 // 4, 1032), "c3s": z(V * n, h8 + 3, w8 + 3, 256),
                "slots": z(40)}
        return b

    def _side_stream(self, dev, main):
        """One side stream per stream the model is called on (FramePipeline calls it on two)."""
        key = (str(dev), main.cuda_stream)
        st = self._sides.get(key)
        if st is None:
            st = self._sides[key] = torch.cuda.Stream(device=dev)
        return st

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, image_key, images_source, intrinsics_key, intrinsics_source, source_to_key):
        """image_key (n,3,H,W), images_source V x (n,3,H,W) (all one size, H and W multiples of 64) -> the decoder's dict of
        predictions (dispnet_decoder.py:126-138)."""
        m, w = self.model, self._prepare()
        dev = image_key.device
        n, _, H, W = image_key.shape
        V = len(images_source)
        h8, w8 = H // 8, W // 8
        S = m.SWEEP["num_sampling_points"]
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        bufs = self._buffers(n, H, W, V, dev)
        slots = bufs["slots"].zero_()  # max-|x| slots, raised atomically by the producers
        _next = [0]

        def slot():
            _next[0] += 1
            return slots[_next[0] - 1:_next[0]]

        def layer(name, x, ax, **kw):
            """One convolution into a buffer of its own -> (output, its max-|x| slot)."""
            a = slot()
            return ops.conv2d_split(x, ax, w[name], out_absmax=a, **kw), a

        # ---- encoder: key view into the decoder's concat buffers, source views as one batch into K1's bordered maps -------
        # The key view's chain (4 launches on ONE image: they fill half the chip) runs on a side stream beside the source views'.
        cat5, cat4 = bufs["cat5"], bufs["cat4"]
        a_cat5, a_cat4, a_c3k, a_merged = slot(), slot(), slot(), slot()
        c3k, merged = e(n, h8, w8, 256), e(n, h8, w8, 288)
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev, main) if self.side_stream else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            a_key = ops.absmax(image_key)
            a_key.record_stream(main)
            ops.conv2d_split(image_key, a_key, w["conv1"], out=cat5[..., :64], out_absmax=a_cat5)
            ops.conv2d_split(cat5[..., :64], a_cat5, w["conv2"], out=cat4[..., :128], out_absmax=a_cat4)
            # (conv2 reads before the decoder adds to the slots: every later writer only raises them, which stays a valid bound)
            ops.conv2d_split(cat4[..., :128], a_cat4, w["conv3"], out=c3k, out_absmax=a_c3k)
            ops.conv2d_split(c3k, a_c3k, w["conv_redir"], out=merged[..., :32], out_absmax=a_merged)
        # (one launch over the batch of source images: per-view launches of this layer measured 2.2x slower, tail effects; the batch is a
        # view, not a copy, when the images are slices of one buffer, as the input adapter leaves them)
        from .models import _as_batch
        src = _as_batch(list(images_source))
        s1, a1 = layer("conv1", src, ops.absmax(src))
        s2, a2 = layer("conv2", s1, a1)
        del s1
        c3s = bufs["c3s"]
        ops.conv2d_split(s2, a2, w["conv3"], out=c3s[:, 1:h8 + 1, 1:w8 + 1, :])
        del s2
        main.wait_stream(side)

        # ---- sweep, learned fusion -------------------------------------------------------------------------------------------------
        inv = m.corr_block.warm(sampling_type="linear_invdepth", device=dev, **m.SWEEP)  # (1,S), cached on the device
        if V == 1:  # LearnedFusion passes a single view through (learned_fusion.py:28-30)
            mask = e(n, h8, w8, 288)[..., 32:]  # the pixel stride of its correlation map
            ops.sweep_corr_nhwc(c3k, [c3s], intrinsics_key, intrinsics_source, source_to_key, inv, [merged[..., 32:]], [mask],
                                corr_absmax=a_merged)
        else:
            corr, mask = e(V * n, h8, w8, S), e(V * n, h8, w8, S)
            views = lambda t: [t[v * n:(v + 1) * n] for v in range(V)]
            a_corr = slot()
            ops.sweep_corr_nhwc(c3k, views(c3s), intrinsics_key, intrinsics_source, source_to_key, inv, views(corr), views(mask),
                                corr_absmax=a_corr)
            mid, a_mid = layer("score3", corr, a_corr, act=2)
            scores = ops.conv2d_split(mid, a_mid, w["score1"], act=0)
            del mid
            ops.fuse_views_nhwc(views(corr), views(mask), views(scores), merged[..., 32:], out_absmax=a_merged)
            del corr, mask

        # ---- cost-volume encoder: the skip outputs go straight into the decoder's concat buffers ---------------------------------
        cat3, cat2, cat1 = bufs["cat3"], bufs["cat2"], bufs["cat1"]
        a_cat3, a_cat2, a_cat1 = slot(), slot(), slot()
        ops.conv2d_split(merged, a_merged, w["conv3_1"], out=cat3[..., :256], out_absmax=a_cat3)
        c4, a_c4 = layer("conv4", cat3[..., :256], a_cat3)
        ops.conv2d_split(c4, a_c4, w["conv4_1"], out=cat2[..., :512], out_absmax=a_cat2)
        c5, a_c5 = layer("conv5", cat2[..., :512], a_cat2)
        ops.conv2d_split(c5, a_c5, w["conv5_1"], out=cat1[..., :512], out_absmax=a_cat1)
        c6, a_c6 = layer("conv6", cat1[..., :512], a_cat1)
        feat, a_feat = layer("conv6_1", c6, a_c6)

        # ---- decoder (dispnet_decoder.py:109-124) ---------------------------------------------------------------------------------------
        preds = {}

        def head(lvl, feat, a_feat):
            raw = ops.conv2d_split(feat, a_feat, w[f"pred_{lvl}"], act=0, planar_out=True)  # (n,2,h,w)
            pred, ent = ops.dispnet_head(raw)
            mean, log_b = pred[:, 0:1], pred[:, 1:2]
            preds.setdefault("invdepth_uncertainties_all", []).append(ent)
            preds.setdefault("invdepth_log_bs_all", []).append(log_b)
            preds.setdefault("invdepths_all", []).append(mean)
            preds["invdepth_uncertainty"], preds["invdepth_log_b"], preds["invdepth"] = ent, log_b, mean
            return pred

        pred = head(0, feat, a_feat)
        for lvl, (buf, a_buf, cs, cu) in enumerate([(cat1, a_cat1, 512, 512), (cat2, a_cat2, 512, 256), (cat3, a_cat3, 256, 128),
                                                   (cat4, a_cat4, 128, 64), (cat5, a_cat5, 64, 32)], start=1):
            ops.conv2d_split(feat, a_feat, w[f"deconv_{lvl}"], out=buf[..., cs:cs + cu], out_absmax=a_buf)
            ops.upsample2x_into(pred, buf[..., cs + cu:cs + cu + 2], out_absmax=a_buf)
            feat, a_feat = layer(f"rfeat{lvl}", buf, a_buf)
            pred = head(lvl, feat, a_feat)
        return preds
