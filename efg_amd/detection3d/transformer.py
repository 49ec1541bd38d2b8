"""Voxel-DETR / ConQueR transformer ($CQ/transformer.py): box-attention encoder over the BEV
tokens, top-k proposal selection, iterative-refinement decoder, momentum GT decoder.

Module and parameter names follow the reference (encoder.layers.N.self_attn..., decoder.layers.N...,
decoder.detection_head, proposal_head, decoder_gt) so state dicts are interchangeable."""
import contextlib
import os

import torch
from torch import nn
from torch.autograd.profiler import record_function
from torch.nn import functional as F

from ..operators import attention
from ..operators.det_loss import box_refine
from ..operators.layernorm import add_layer_norm
from ..operators.linear import Linear, linear, self_attention_in_proj
from .box_attention import Box3dAttention
from .losses import PaddedTargets
from . import encoder_layer as _encoder_layer
from .utils import MLP, flatten_with_shape, get_clones, inverse_sigmoid


def _with_pos(tensor, pos):
    return tensor if pos is None else tensor + pos


class TransformerEncoderLayer(nn.Module):
    """$CQ/transformer.py:206-243: box self-attention (no rotation) + FFN, post-norm."""

    def __init__(self, d_model, nhead, nlevel, dim_feedforward, dropout, activation="relu"):
        super().__init__()
        self.self_attn = Box3dAttention(d_model, nlevel, nhead, with_rotation=False)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        assert activation == "relu"
        self.activation = F.relu

    def forward(self, src, pos, src_shape, src_start_idx, ref_windows):
        if _encoder_layer.usable(self, src, pos, ref_windows):
            # each half as one autograd node: the same forward kernels, the input gradients of its projections accumulated
            # by their GEMMs instead of by addition kernels (detection3d/encoder_layer.py; EFG_FUSED_ENCODER=0: this form)
            return _encoder_layer.forward(self, src, pos, src_shape, src_start_idx, ref_windows)
        src2 = self.self_attn(_with_pos(src, pos), src, src_shape, None, src_start_idx, None, ref_windows)[0]
        src = add_layer_norm(src, self.dropout1(src2), self.norm1)
        hidden = linear(src, self.linear1.weight, self.linear1.bias, relu=True)  # activation(linear1(src))
        src2 = self.linear2(self.dropout(hidden))
        return add_layer_norm(src, self.dropout2(src2), self.norm2)


class TransformerEncoder(nn.Module):
    def __init__(self, d_model, encoder_layer, num_layers):
        super().__init__()
        self.layers = get_clones(encoder_layer, num_layers)

    def forward(self, src, pos, src_shape, src_start_idx, ref_windows):
        output = src
        for layer in self.layers:
            output = layer(output, pos, src_shape, src_start_idx, ref_windows)
        return output


class TransformerDecoderLayer(nn.Module):
    """$CQ/transformer.py:258-317: MHA self-attention (bool attn_mask) -> rotated box cross-attention
    over the encoder memory -> FFN.  Layer 0 REPLACES the query content by pos_embed_layer(ref)
    and runs cross-attention with query_pos=None (:284-289) -- reproduced, not 'fixed'."""

    def __init__(self, d_model, nhead, nlevel, dim_feedforward, dropout, activation="relu"):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = Box3dAttention(d_model, nlevel, nhead, with_rotation=True)
        self.pos_embed_layer = MLP(10, d_model, d_model, 3)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        assert activation == "relu"
        self.activation = F.relu

    def _self_attention(self, qk_in, v_in, attn_mask, attn_bits):
        """self_attn(q = k = qk_in, v = v_in) of the nn.MultiheadAttention module, batch first."""
        mha = self.self_attn
        c, h = mha.embed_dim, mha.num_heads
        if (attention.fused_long(v_in, h) and (attn_mask is None or attn_bits is not None) and mha.dropout == 0.0
                and mha.in_proj_weight is not None):
            # csrc/attention.hip (long / 32-wide-head kernels): q | k from ONE projection of the shared input, the boolean
            # mask as bit rows, one gradient tensor for that projection
            qk, v = self_attention_in_proj(qk_in, v_in, mha.in_proj_weight, mha.in_proj_bias)
            return F.linear(attention.long_self_attention(qk, v, attn_bits, h), mha.out_proj.weight, mha.out_proj.bias)
        q, v = qk_in.transpose(0, 1), v_in.transpose(0, 1)
        # need_weights=False: same output, skips materialising the head-averaged attention map the reference
        # computes and discards ($CQ/transformer.py:295)
        return mha(q, q, v, attn_mask=attn_mask, need_weights=False)[0].transpose(0, 1)

    def forward(self, idx, query, query_pos, memory, memory_shape, memory_start_idx, ref_windows, attn_mask=None,
                attn_bits=None):
        if idx == 0:
            query = self.pos_embed_layer(ref_windows)
            qk_in = query
        elif query_pos is None:
            query_pos = self.pos_embed_layer(ref_windows)
            qk_in = _with_pos(query, query_pos)
        query2 = self._self_attention(qk_in, query, attn_mask, attn_bits)
        query = add_layer_norm(query, self.dropout1(query2), self.norm1)
        query2 = self.multihead_attn(_with_pos(query, query_pos), memory, memory_shape, None, memory_start_idx, None,
                                     ref_windows[..., :7])[0]
        query = add_layer_norm(query, self.dropout2(query2), self.norm2)
        hidden = linear(query, self.linear1.weight, self.linear1.bias, relu=True)   # activation(linear1(query))
        query2 = self.linear2(self.dropout(hidden))
        return add_layer_norm(query, self.dropout3(query2), self.norm3)


class TransformerDecoder(nn.Module):
    """$CQ/transformer.py:320-343; `detection_head` is attached by the model (voxel_detr.py:79-85)."""

    def __init__(self, d_model, decoder_layer, num_layers):
        super().__init__()
        self.layers = get_clones(decoder_layer, num_layers)

    def forward(self, query, query_pos, memory, memory_shape, memory_start_idx, ref_windows, attn_mask=None):
        output = query
        intermediate, intermediate_ref_windows = [], []
        attn_bits = None   # the boolean mask as bit rows, packed once for all layers (operators/attention.py)
        if (attn_mask is not None and attn_mask.is_cuda and attn_mask.dtype == torch.bool and attn_mask.dim() == 2
                and attn_mask.shape[0] == attn_mask.shape[1] and os.environ.get("EFG_ATTENTION", "1") != "0"):
            attn_bits = attention.pack_mask(attn_mask)
        for idx, layer in enumerate(self.layers):
            output = layer(idx, output, query_pos, memory, memory_shape, memory_start_idx, ref_windows, attn_mask,
                           attn_bits)
            new_ref_logits, new_ref_windows = self.detection_head(output, ref_windows[..., :7], idx)
            ref_windows = torch.cat((new_ref_windows.detach(), new_ref_logits.sigmoid().detach()), dim=-1)
            intermediate.append(output)
            intermediate_ref_windows.append(new_ref_windows)
        return torch.stack(intermediate), torch.stack(intermediate_ref_windows)


class Transformer(nn.Module):
    """$CQ/transformer.py:10-203."""

    def __init__(self, d_model=256, nhead=8, nlevel=4, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=1024, dropout=0.1, activation="relu", num_queries=300, num_classes=3, mom=0.999):
        super().__init__()
        self.num_queries, self.num_classes, self.m = num_queries, num_classes, mom
        encoder_layer = TransformerEncoderLayer(d_model, nhead, nlevel, dim_feedforward, dropout, activation)
        self.encoder = TransformerEncoder(d_model, encoder_layer, num_encoder_layers)
        decoder_layer = TransformerDecoderLayer(d_model, nhead, nlevel, dim_feedforward, dropout, activation)
        self.decoder = TransformerDecoder(d_model, decoder_layer, num_decoder_layers)

    def _create_ref_windows(self, tensor_list):
        """Fixed anchors per BEV token: (x, y, z=0.5, l=w=0.025, h=0.5, angle=0) (:36-58)."""
        device = tensor_list[0].device
        key = (tuple(tuple(t.shape[0:1]) + tuple(t.shape[2:]) for t in tensor_list), device)
        if getattr(self, "_ref_cache_key", None) == key:  # constants of the BEV shape: build once
            return self._ref_cache
        ref_windows = []
        for tensor in tensor_list:
            B, _, H, W = tensor.shape
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W - 0.5, W, dtype=torch.float32, device=device),
                                          indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / H
            ref_x = ref_x.reshape(-1)[None] / W
            ref_xy = torch.stack((ref_x, ref_y), -1)
            ref_wh = torch.ones_like(ref_xy) * 0.025
            ph = torch.zeros_like(ref_xy)[..., :1]
            ref_windows.append(torch.cat((ref_xy, ph + 0.5, ref_wh, ph + 0.5, ph), -1).expand(B, -1, -1))
        self._ref_cache_key, self._ref_cache = key, torch.cat(ref_windows, dim=1).contiguous()
        return self._ref_cache

    def _select_proposals(self, probs):
        """(scores, token indexes) of the num_queries best proposals per scene, unsorted ($CQ/transformer.py:65)."""
        if probs.is_cuda and probs.dtype == torch.float32 and probs.dim() == 2 and os.environ.get("EFG_TOPK", "1") != "0":
            # one launch and a FIXED tie rule (lowest indices of the values equal to the k-th): csrc/topk.hip
            from ..operators.det_loss import topk_unsorted

            return topk_unsorted(probs, self.num_queries)
        return torch.topk(probs, self.num_queries, dim=1, sorted=False)

    def _get_enc_proposals(self, enc_embed, ref_windows):
        """top-k (unsorted, :65) proposals of the 1-class proposal head, detached (:60-81).

        The reference evaluates the proposal head over every token here, detaches the result, and evaluates it a
        SECOND time in the loss (voxel_detr.py:198).  Here the class MLP runs once over all tokens (kept, with its
        graph, for the encoder classification loss) and the box MLP only on the top-k tokens -- the only boxes
        anything reads (proposals, matcher, box loss): per-token MLPs commute with the gather."""
        head = self.proposal_head
        out_logits = head.class_embed[0](enc_embed)
        out_probs = out_logits[..., 0].sigmoid()
        topk_probs, indexes = self._select_proposals(out_probs.detach())
        topk_probs, indexes = topk_probs.unsqueeze(-1), indexes.unsqueeze(-1)
        emb_k = torch.gather(enc_embed, 1, indexes.expand(-1, -1, enc_embed.shape[-1]))
        ref_k = torch.gather(ref_windows, 1, indexes.expand(-1, -1, ref_windows.shape[-1]))
        boxes_k = box_refine(head.bbox_embed[0](emb_k), ref_k)   # (delta + inverse_sigmoid(ref_k)).sigmoid()
        # (plain instance attribute, set past nn.Module.__setattr__: its walk over the parameter / buffer / module tables
        # costs 50-80 us per assignment on the host)
        self.__dict__["enc_outputs"] = {"pred_logits": out_logits, "topk_boxes": boxes_k, "topk_indexes": indexes}
        out_ref_windows = torch.cat((boxes_k.detach(), topk_probs.detach().expand(-1, -1, 3)), dim=-1)
        return None, None, out_ref_windows, indexes

    def _gt_stream(self, device):
        """Side stream of the momentum decoder (one per process and device, efg_amd/streams.py); None on the CPU or with
        EFG_GT_STREAM=0."""
        if device.type != "cuda" or os.environ.get("EFG_GT_STREAM", "1") == "0":
            return None
        from ..streams import side_stream

        return side_stream(device, "gt_decoder")

    def _launch_gt_decoder(self, memory, src_shape, src_start_index, targets, noised_gt_proposals):
        """EMA update + momentum decoder over the GT boxes (and their positive-noised copies), $CQ/transformer.py:146-200.

        The pass needs the encoder memory and the targets only, and nothing before the contrastive loss needs its result:
        it is issued right after the encoder on its OWN stream, so that its ~160 launches (one graph replay) run beside
        the proposal head and the decoder -- both are chains of small kernels that leave the device mostly idle -- instead
        of after them.  Returns what _join_gt_decoder needs."""
        main = torch.cuda.current_stream(memory.device) if memory.is_cuda else None
        side = self._gt_stream(memory.device)
        if side is not None:
            side.wait_stream(main)   # the memory, the noised proposals, last step's optimizer update of the decoder
        batch_size = len(targets)
        per_gt_num = [tgt["gt_boxes"].shape[0] for tgt in targets]
        max_gt_num = max(per_gt_num)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            if isinstance(targets, PaddedTargets):  # batched: padded rows are zeroed by the validity mask
                valid = (torch.arange(max_gt_num)[None, :] < torch.tensor(per_gt_num)[:, None]).to(
                    memory.device, non_blocking=True)
                gt_with_score = torch.cat((targets.boxes[:, :max_gt_num],
                                           F.one_hot(targets.labels[:, :max_gt_num], num_classes=self.num_classes)
                                           .to(memory.dtype)), dim=-1) * valid[..., None]
            else:
                gt_with_score = memory.new_zeros(batch_size, max_gt_num, 10)
                for bi in range(batch_size):
                    gt_with_score[bi, : per_gt_num[bi], :7] = targets[bi]["gt_boxes"]
                    gt_with_score[bi, : per_gt_num[bi], 7:] = F.one_hot(targets[bi]["labels"],
                                                                        num_classes=self.num_classes)
            with torch.no_grad(), record_function("efg::gt_decoder"):
                self._momentum_update_gt_decoder()
                if noised_gt_proposals is not None:
                    dn_group_num = noised_gt_proposals.shape[1] // (max_gt_num * 2)
                    pos_noised = torch.cat([noised_gt_proposals[:, pi * max_gt_num:(pi + 1) * max_gt_num]
                                            for pi in range(0, dn_group_num * 2, 2)], dim=1)
                    gt_proposals = torch.cat((gt_with_score, pos_noised), dim=1)
                    n = (dn_group_num + 1) * max_gt_num
                    grp = torch.arange(n, device=memory.device) // max_gt_num
                    gt_attn_mask = grp[:, None] != grp[None, :]  # groups see only themselves
                else:
                    gt_proposals, gt_attn_mask = gt_with_score, None
                hs_gt, inter_references_gt = self._run_gt_decoder(memory, src_shape, src_start_index,
                                                                  gt_proposals, gt_attn_mask)
        return hs_gt, inter_references_gt, gt_proposals, side, main

    @staticmethod
    def _join_gt_decoder(run):
        hs_gt, inter_references_gt, gt_proposals, side, main = run
        if side is not None:
            main.wait_stream(side)
            for t in (hs_gt, inter_references_gt, gt_proposals):   # allocated on the side stream, read on the main one
                t.record_stream(main)
        return hs_gt, inter_references_gt, gt_proposals

    def _run_gt_decoder(self, memory, src_shape, src_start_index, gt_proposals, gt_attn_mask):
        """The momentum decoder is a no-grad pass of ~160 small kernels over a few hundred queries: 2.5-3.8 ms of
        host launch work for 1.2 ms of device time, on a step whose host side is within 10 % of its device side.
        On the GPU it is captured once per input shape into a HIP graph and replayed (inputs copied into the
        graph's static buffers; the parameters are read in place, so the EMA update before it stays visible).
        EFG_GT_GRAPH=0, a CPU model or a failed capture run it eagerly."""
        if (not memory.is_cuda or os.environ.get("EFG_GT_GRAPH", "1") == "0" or getattr(self, "_gt_graph_off", False)
                or torch.is_grad_enabled()):
            return self.decoder_gt(None, None, memory, src_shape, src_start_index, gt_proposals, gt_attn_mask)
        cache = self.__dict__.setdefault("_gt_graphs", {})
        key = (tuple(memory.shape), tuple(src_shape.shape), tuple(gt_proposals.shape),
               None if gt_attn_mask is None else tuple(gt_attn_mask.shape), memory.device.index)
        ent = cache.get(key)
        live = (memory, src_shape, src_start_index, gt_proposals, gt_attn_mask)
        stats = self.__dict__.setdefault("_gt_graph_stats", [0, 0])  # hits, captures
        if ent is not None:
            stats[0] += 1
        elif stats[1] >= 16 and stats[1] > stats[0]:
            # the padded GT count changes faster than shapes repeat: capturing (3 passes + instantiation) costs more
            # than it saves -- stay eager
            self._gt_graph_off = True
            return self.decoder_gt(None, None, memory, src_shape, src_start_index, gt_proposals, gt_attn_mask)
        if ent is None:
            from .. import _prof
            stats[1] += 1
            if len(cache) >= 8:  # ragged GT counts: keep the pool of captured shapes bounded
                cache.pop(next(iter(cache)))
            static = [None if t is None else t.clone() for t in live]
            was_on = _prof.active()
            _prof._enabled = False  # event records do not belong into the graph (enable() would clear the records)
            try:
                from ..hipgraph import capture
                graph, out = capture(lambda: self.decoder_gt(None, None, static[0], static[1], static[2], static[3],
                                                             static[4]), memory.device)
            except Exception as exc:  # noqa: BLE001 -- any capture problem: run eagerly from now on, say so once
                _prof._enabled = was_on
                if os.environ.get("EFG_GT_GRAPH_STRICT", "0") == "1":  # tests/test_gt_graph_gpu.py: a failed capture is a failure
                    raise
                import warnings
                warnings.warn("efg_amd: HIP-graph capture of the momentum decoder failed (%s); running it eagerly" % exc)
                self._gt_graph_off = True
                return self.decoder_gt(None, None, memory, src_shape, src_start_index, gt_proposals, gt_attn_mask)
            _prof._enabled = was_on
            ent = cache[key] = (graph, static, out)
        graph, static, out = ent
        for dst, src in zip(static, live):
            if dst is not None:
                dst.copy_(src)
        graph.replay()
        return out

    @torch.no_grad()
    def _momentum_update_gt_decoder(self):
        # the two parameter lists are walked once (module traversal: ~1 ms of host time per step for the two decoders);
        # nn.Module.to() / load_state_dict() keep the Parameter objects, and a changed count rebuilds the lists
        lists = self.__dict__.get("_ema_lists")
        if lists is None or lists[2] != (len(self.decoder._modules), id(self.decoder), id(self.decoder_gt)):
            qs, ks = list(self.decoder.parameters()), list(self.decoder_gt.parameters())
            lists = self.__dict__["_ema_lists"] = (qs, ks, (len(self.decoder._modules), id(self.decoder), id(self.decoder_gt)))
        qs, ks = lists[0], lists[1]
        # param_k = param_k * m + param_q * (1 - m) (:84-89), one fused multi-tensor pass
        torch._foreach_mul_(ks, self.m)
        torch._foreach_add_(ks, qs, alpha=1.0 - self.m)

    def forward(self, src, pos, noised_gt_box=None, noised_gt_onehot=None, attn_mask=None, targets=None):
        assert pos is not None, "position encoding is required!"
        src_anchors = self._create_ref_windows(src)
        src, src_shape = flatten_with_shape(src)
        if len(pos) == 1:
            src_pos = pos[0].flatten(2).transpose(1, 2).contiguous()  # cached channels-last: a view, no copy
        else:
            src_pos = torch.cat([pe.flatten(2).transpose(1, 2) for pe in pos], dim=1)
        src_start_index = torch.cat([src_shape.new_zeros(1), src_shape.prod(1).cumsum(0)[:-1]])
        with record_function("efg::encoder"):
            memory = self.encoder(src, src_pos, src_shape, src_start_index, src_anchors)
        with record_function("efg::proposals"):
            query_embed, query_pos, topk_proposals, topk_indexes = self._get_enc_proposals(memory, src_anchors)
        if noised_gt_box is not None:
            noised_gt_proposals = torch.cat((noised_gt_box, noised_gt_onehot), dim=-1)
            topk_proposals = torch.cat((noised_gt_proposals, topk_proposals), dim=1)
        init_reference_out = topk_proposals[..., :7]
        gt_run = None
        if targets is not None:  # momentum GT decoder pass (:146-200): issued BEFORE the decoder, beside it (see _launch_gt_decoder)
            gt_run = self._launch_gt_decoder(memory, src_shape, src_start_index, targets,
                                             noised_gt_proposals if noised_gt_box is not None else None)
        with record_function("efg::decoder"):
            hs, inter_references = self.decoder(query_embed, query_pos, memory, src_shape, src_start_index,
                                                topk_proposals, attn_mask)
        if gt_run is not None:
            hs_gt, inter_references_gt, gt_proposals = self._join_gt_decoder(gt_run)
            init_reference_out = torch.cat((init_reference_out, gt_proposals[..., :7]), dim=1)
            hs = torch.cat((hs, hs_gt), dim=2)
            inter_references = torch.cat((inter_references, inter_references_gt), dim=2)
        return hs, init_reference_out, inter_references, memory, src_anchors, topk_indexes
