"""Contrastive denoising queries ($CQ/cdn.py:5-139): 2*dn_number noised copies of every GT box
(positive copy: noise in [0,1) * half size, negative copy: [1,2)), label flips, group attention mask."""
import torch
from torch.nn import functional as F


def prepare_for_cdn(dn_args, training, num_queries, num_classes, hidden_dim, label_enc, generator=None,
                    with_mask=True):
    if not training:
        return None, None, None, None
    targets, dn_number, label_noise_ratio, box_noise_scale = dn_args
    dev = targets[0]["gt_boxes"].device
    known_num = [int(t["labels"].numel()) for t in targets]
    batch_size = len(targets)
    labels = torch.cat([t["labels"] for t in targets])
    boxes = torch.cat([t["gt_boxes"] for t in targets])
    batch_idx = torch.cat([torch.full_like(t["labels"].long(), i) for i, t in enumerate(targets)])
    known_labels = labels.repeat(2 * dn_number, 1).view(-1)
    known_bid = batch_idx.repeat(2 * dn_number, 1).view(-1)
    known_bboxs = boxes.repeat(2 * dn_number, 1)
    known_labels_expaned = known_labels.clone()
    known_bbox_expand = known_bboxs.clone()

    gdev = generator.device if generator is not None else dev  # a CPU generator gives device-independent noise

    def rand_like(t):
        return torch.rand(t.shape, dtype=torch.float32, device=gdev, generator=generator).to(dev)

    def randint(lo, hi, shape, dtype=torch.int64):
        return torch.randint(lo, hi, tuple(shape), device=gdev, generator=generator, dtype=dtype).to(dev)

    if label_noise_ratio > 0:
        p = rand_like(known_labels_expaned)
        chosen = torch.nonzero(p < (label_noise_ratio * 0.5)).view(-1)
        new_label = randint(0, num_classes, chosen.shape, known_labels_expaned.dtype)
        known_labels_expaned.scatter_(0, chosen, new_label)
    single_pad = int(max(known_num))
    pad_size = int(single_pad * 2 * dn_number)
    nb = len(boxes)
    positive_idx = (torch.arange(nb, device=dev).unsqueeze(0).repeat(dn_number, 1) +
                    (torch.arange(dn_number, device=dev) * nb * 2).unsqueeze(1)).flatten()
    negative_idx = positive_idx + nb
    if box_noise_scale > 0:
        corners = torch.zeros_like(known_bboxs)
        corners[:, :3] = known_bboxs[:, :3] - known_bboxs[:, 3:6] / 2
        corners[:, 3:6] = known_bboxs[:, :3] + known_bboxs[:, 3:6] / 2
        corners[:, 6:] = known_bboxs[:, 6:]
        diff = torch.zeros_like(known_bboxs)
        diff[:, :3] = known_bboxs[:, 3:6] / 2
        diff[:, 3:6] = known_bboxs[:, 3:6] / 2
        diff[:, 6:] = 0.1
        rand_sign = randint(0, 2, known_bboxs.shape).float() * 2.0 - 1.0
        rand_part = rand_like(known_bboxs)
        rand_part[negative_idx] += 1.0
        rand_part *= rand_sign
        corners = (corners + rand_part * diff * box_noise_scale).clamp(min=0.0, max=1.0)
        known_bbox_expand[:, :3] = (corners[:, :3] + corners[:, 3:6]) / 2
        known_bbox_expand[:, 3:6] = corners[:, 3:6] - corners[:, :3]
        known_bbox_expand[:, 6:] = corners[:, 6:]
    input_label_embed = F.one_hot(known_labels_expaned.long(), num_classes=num_classes).float()
    input_query_label = torch.zeros(batch_size, pad_size, num_classes, device=dev)
    input_query_bbox = torch.zeros(batch_size, pad_size, 7, device=dev)
    if len(known_bid):
        within = torch.cat([torch.arange(n, device=dev) for n in known_num])
        map_known = torch.cat([within + single_pad * i for i in range(2 * dn_number)]).long()
        input_query_label[(known_bid.long(), map_known)] = input_label_embed
        input_query_bbox[(known_bid.long(), map_known)] = known_bbox_expand
    mask = dn_attn_mask(pad_size, single_pad, dn_number, num_queries, dev) if with_mask else None
    return input_query_label, input_query_bbox, mask, {"pad_size": pad_size, "num_dn_group": dn_number}


def dn_attn_mask(pad_size, single_pad, dn_number, num_queries, device):
    """True = blocked.  Matching queries cannot see DN queries and vice versa; DN groups (pos+neg
    pair of width 2*single_pad) only see themselves ($CQ/cdn.py:98-112)."""
    tgt_size = pad_size + num_queries
    grp = torch.arange(tgt_size, device=device)
    grp = torch.where(grp < pad_size, grp // max(2 * single_pad, 1), torch.full_like(grp, dn_number))
    return grp[:, None] != grp[None, :]


def dn_post_process(outputs_class, outputs_coord, dn_meta, aux_loss, _set_aux_loss):
    """Split the DN part off the decoder outputs ($CQ/cdn.py:122-139)."""
    if dn_meta and dn_meta["pad_size"] > 0:
        pad = dn_meta["pad_size"]
        known_class, known_coord = outputs_class[:, :, :pad, :], outputs_coord[:, :, :pad, :]
        outputs_class, outputs_coord = outputs_class[:, :, pad:, :], outputs_coord[:, :, pad:, :]
        out = {"pred_logits": known_class[-1], "pred_boxes": known_coord[-1]}
        if aux_loss:
            out["aux_outputs"] = _set_aux_loss(known_class[:-1], known_coord[:-1])
        dn_meta["output_known_lbs_bboxes"] = out
    return outputs_class, outputs_coord
