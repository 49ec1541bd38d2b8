"""Learned blocks of TrajectoryFormer: the point / hypothesis encoders, the trajectory PointNet and the motion
polyline encoder.

Counterparts of $TF/transformer.py, $TF/pointnet.py and $TF/modules/blocks.py.  Module and parameter names equal the
reference's, so its checkpoints load (`encoder_fg.layers.0.point_attn.in_proj_weight`, `seqboxembed.feat.conv1.weight`,
`velboxembed.pre_mlps.0.weight`, ...).  The arithmetic is the reference's; the layout is not: attention is evaluated
batch-first through `scaled_dot_product_attention` on the [batch, head, token, dim] view of one fused QKV GEMM --
no [token, batch] permutes and no materialised head-averaged attention map, which the reference computes
(nn.MultiheadAttention's default need_weights=True) and discards.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..operators import attention
from ..operators.batchnorm import bn_act
from ..operators.layernorm import add_layer_norm  # norm(x + r): one fused HIP pass each way on the GPU
from ..operators.linear import linear  # F.linear; on long GPU matrices: split weight gradient + HIP bias gradient


class MLP(nn.Module):
    """Linear/ReLU stack, no activation after the last layer (blocks.py:5-19)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        widths = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        for layer in self.layers[:-1]:
            x = linear(x, layer.weight, layer.bias, relu=True)
        return linear(x, self.layers[-1].weight, self.layers[-1].bias)


def attend(mha, query, memory):
    """softmax(Q K^T / sqrt(d)) V with the parameters of an nn.MultiheadAttention, batch-first:
    query [B, Lq, C], memory [B, Lk, C] (keys = values) -> [B, Lq, C].  The core runs csrc/attention.hip on the GPU for
    up to 128 tokens and 64-wide heads (operators/attention.py), SDPA otherwise."""
    c, h = mha.embed_dim, mha.num_heads
    if query is memory:
        out = attention.self_attention_qkv(linear(query, mha.in_proj_weight, mha.in_proj_bias), h)
    else:
        out = attention.cross_attention_kv(linear(query, mha.in_proj_weight[:c], mha.in_proj_bias[:c]),
                                           linear(memory, mha.in_proj_weight[c:], mha.in_proj_bias[c:]), h)
    return linear(out, mha.out_proj.weight, mha.out_proj.bias)


class TransformerEncoderLayer(nn.Module):
    """Point encoder layer (transformer.py:44-92): self-attention over a ROI's points, then a learned token summarises
    them.  The points and the token share one FFN and one pair of LayerNorms, as in the reference."""

    def __init__(self, config, d_model, nhead, dim_feedforward=2048, dropout=0):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.point_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = F.relu

    def _mix(self, x, update):
        x = add_layer_norm(x, self.dropout1(update), self.norm1)
        hidden = self.dropout(linear(x, self.linear1.weight, self.linear1.bias, relu=True))   # activation = ReLU
        return add_layer_norm(x, self.dropout2(linear(hidden, self.linear2.weight, self.linear2.bias)), self.norm2)

    def forward(self, token, src, pos=None):
        src = self._mix(src, attend(self.point_attn, src, src))
        token = self._mix(token, attend(self.self_attn, token, src))
        return src, token


class TransformerEncoder(nn.Module):
    """Stack of point layers; returns the summary token after every layer (transformer.py:5-21)."""

    def __init__(self, encoder_layer, num_layers, norm=None, config=None):
        super().__init__()
        self.layers = nn.ModuleList(encoder_layer)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, token, src, pos=None):
        tokens = []
        for layer in self.layers:
            src, token = layer(token, src, pos=pos)
            tokens.append(token)
        return tokens


class FFN(nn.Module):
    """Residual add + LayerNorm + FFN + LayerNorm (transformer.py:141-170)."""

    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, dout=None, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]

    def forward(self, tgt, tgt_input):
        tgt = add_layer_norm(tgt, self.dropout2(tgt_input), self.norm2)
        if self.activation is F.relu:
            hidden = linear(tgt, self.linear1.weight, self.linear1.bias, relu=True)
        else:
            hidden = self.activation(linear(tgt, self.linear1.weight, self.linear1.bias))
        return add_layer_norm(tgt, self.dropout3(linear(self.dropout(hidden), self.linear2.weight, self.linear2.bias)),
                              self.norm3)


class TransformerEncoderLayerGlobalLocal(nn.Module):
    """Hypothesis encoder layer (transformer.py:95-138): attention over all hypotheses of a scene, then within the
    hypotheses of each track."""

    def __init__(self, config, d_model, nhead, dim_feedforward=2048, dropout=0):
        super().__init__()
        self.global_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.local_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.ffn1 = FFN(d_model, dim_feedforward)
        self.ffn2 = FFN(d_model, dim_feedforward)
        self.activation = F.relu

    def forward(self, src):
        bs, num_track, candi, c = src.shape
        scene = src.reshape(bs, num_track * candi, c)
        scene = self.ffn1(scene, attend(self.global_attn, scene, scene))
        track = scene.reshape(bs * num_track, candi, c)
        track = self.ffn2(track, attend(self.local_attn, track, track))
        return track.reshape(bs, num_track, candi, c)


class TransformerEncoderGlobalLocal(nn.Module):
    """Stack of hypothesis layers; returns every layer's output (transformer.py:24-41)."""

    def __init__(self, encoder_layer, num_layers, norm=None, config=None):
        super().__init__()
        self.layers = nn.ModuleList(encoder_layer)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src):
        outputs = []
        for layer in self.layers:
            src = layer(src)
            outputs.append(src)
        return outputs


class PointNetfeat(nn.Module):
    """Four 1x1 conv + BN stages, max over the sequence (pointnet.py:136-160).  Kernel-size-1 convolutions are
    row-wise linear maps: the [S, C, T] input is handled as S*T rows through four GEMMs (BatchNorm over rows = over
    (S, T) per channel, the same statistics), which keeps MIOpen's convolution solver search -- triggered anew for
    every track count under `cudnn.benchmark` -- out of the step.  Parameters keep their Conv1d shapes [Cout, Cin, 1]."""

    def __init__(self, input_dim, x=1, outchannel=512):
        super().__init__()
        self.output_channel = 256 if outchannel == 256 else 512 * x
        widths = [input_dim, 64 * x, 128 * x, 256 * x, self.output_channel]
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), nn.Conv1d(widths[i], widths[i + 1], 1))
        for i in range(4):
            setattr(self, "bn%d" % (i + 1), nn.BatchNorm1d(widths[i + 1]))

    def _stage(self, rows, i, relu):
        conv = getattr(self, "conv%d" % i)   # norm (+ ReLU) in one fused pass each way on the GPU (operators/batchnorm.py)
        return bn_act(F.linear(rows, conv.weight.squeeze(-1), conv.bias), getattr(self, "bn%d" % i), relu=relu)

    def forward_rows(self, rows, steps):
        """rows [S * steps, C] (sequence-major) -> (pooled [S, out], per-step [S, steps, out])."""
        for i in (1, 2, 3):
            rows = self._stage(rows, i, True)
        per_step = self._stage(rows, 4, False).view(-1, steps, self.output_channel)
        return per_step.max(dim=1)[0], per_step

    def forward(self, x):
        s, c, t = x.shape
        pooled, per_step = self.forward_rows(x.transpose(1, 2).reshape(s * t, c), t)
        return pooled, per_step.transpose(1, 2)


class PointNet(nn.Module):
    """Trajectory-box embedding (pointnet.py:7-58): BN over the 8 box channels, PointNetfeat over the frames, two
    FC + BN.  The three regression heads (`fc_s*`, `fc_ce*`, `fc_hr*`) exist for checkpoint compatibility; the
    tracker never evaluates them."""

    def __init__(self, input_dim, joint_feat=False, channels=None):
        super().__init__()
        if joint_feat:
            raise NotImplementedError("joint_feat=True is not used by TrajectoryFormer")
        self.joint_feat = joint_feat
        self.feat = PointNetfeat(input_dim, 1)
        self.fc1 = nn.Linear(512, 256)
        self.fc2 = nn.Linear(256, channels)
        self.pre_bn = nn.BatchNorm1d(input_dim)
        self.bn1 = nn.BatchNorm1d(256)
        self.bn2 = nn.BatchNorm1d(channels)
        self.relu = nn.ReLU()
        self.fc_s1 = nn.Linear(channels, 256)
        self.fc_s2 = nn.Linear(256, 3, bias=False)
        self.fc_ce1 = nn.Linear(channels, 256)
        self.fc_ce2 = nn.Linear(256, 3, bias=False)
        self.fc_hr1 = nn.Linear(channels, 256)
        self.fc_hr2 = nn.Linear(256, 1, bias=False)
        self.init_weights()

    def forward(self, x, feat=None):
        """x [S, C, T] -> (embedding [S, channels], per-step features [S, 512, T])."""
        s, c, t = x.shape
        pooled, per_step = self.feat.forward_rows(self.pre_bn(x.transpose(1, 2).reshape(s * t, c)), t)
        hidden = bn_act(self.fc1(pooled), self.bn1, relu=True)
        return bn_act(self.fc2(hidden), self.bn2, relu=True), per_step.transpose(1, 2)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)


def build_mlps(c_in, mlp_channels=None, ret_before_act=False, without_norm=False):
    """Linear(+BN)+ReLU stack, optionally ending in a bare Linear (pointnet.py:163-186)."""
    layers = []
    for k, width in enumerate(mlp_channels):
        last = k + 1 == len(mlp_channels)
        if last and ret_before_act:
            layers.append(nn.Linear(c_in, width, bias=True))
        elif without_norm:
            layers += [nn.Linear(c_in, width, bias=True), nn.ReLU()]
        else:
            layers += [nn.Linear(c_in, width, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
        c_in = width
    return nn.Sequential(*layers)


class MotionEncoder(nn.Module):
    """Polyline encoder of the (pre-trained, frozen) motion model (pointnet.py:61-133): per-step MLP, max over the
    steps, concat, MLP, max, output MLP.  Only valid steps / polylines go through the MLPs (their BatchNorm statistics
    are over valid entries), so the gathers stay; they are index_select / index_copy on a precomputed index instead
    of three boolean-mask round trips."""

    def __init__(self, in_channels, hidden_dim, num_layers=3, num_pre_layers=1, out_channels=None):
        super().__init__()
        self.pre_mlps = build_mlps(in_channels, [hidden_dim] * num_pre_layers)
        self.mlps = build_mlps(hidden_dim * 2, [hidden_dim] * (num_layers - num_pre_layers))
        self.out_mlps = None if out_channels is None else build_mlps(
            hidden_dim, [hidden_dim, hidden_dim, out_channels], ret_before_act=True)

    def forward(self, polylines, polylines_mask):
        b, n, t, c = polylines.shape
        steps = polylines_mask.reshape(-1).nonzero().flatten()

        def scatter_steps(values):
            full = values.new_zeros(b * n * t, values.shape[-1])
            return full.index_copy(0, steps, values).reshape(b, n, t, -1)

        feat = scatter_steps(self.pre_mlps(polylines.reshape(-1, c).index_select(0, steps)))
        pooled = feat.max(dim=2)[0]
        feat = torch.cat((feat, pooled[:, :, None, :].expand(-1, -1, t, -1)), dim=-1)
        feat = scatter_steps(self.mlps(feat.reshape(b * n * t, -1).index_select(0, steps)))
        pooled = feat.max(dim=2)[0]
        if self.out_mlps is None:
            return pooled
        lines = (polylines_mask.sum(dim=-1) > 0).reshape(-1).nonzero().flatten()
        out = self.out_mlps(pooled.reshape(b * n, -1).index_select(0, lines))
        return out.new_zeros(b * n, out.shape[-1]).index_copy(0, lines, out).reshape(b, n, -1)
