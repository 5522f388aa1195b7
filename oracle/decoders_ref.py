"""CPU restatement of the reference decoders' forward pass. TEST INFRASTRUCTURE.

Functional (parameters passed as a dict keyed like the reference's state_dict) so the same
code checks tiny golden cases and full-size runs; gradients come from torch autograd, as in
the reference (loss.backward(), stylenet/train_multitask.py:386).

  factored_lstm_forward  <- DecoderFactoredLSTM.forward / forward_step, stylenet/model.py:115-196
  lstm_forward           <- DecoderRNN.forward / forward_step,          nic/model.py:74-115
  factored_att_forward   <- DecoderFactoredLSTMAtt.forward,             stylenet/model_att.py:238-305
  lstm_att_forward       <- DecoderRNNAtt.forward,                      nic/model_att.py:152-202
The teacher-forcing decisions are an explicit list (one bool per time step); callers draw them
with `random.random() < ratio` in step order, which is what stylenet/model.py:181 does.
"""
import torch
import torch.nn.functional as Fn

MODE_S = {
    "factual": ["S_fi", "S_ff", "S_fo", "S_fc"],
    "happy": ["S_happy_i", "S_happy_f", "S_happy_o", "S_happy_c"],
    "sad": ["S_sad_i", "S_sad_f", "S_sad_o", "S_sad_c"],
    "angry": ["S_angry_i", "S_angry_f", "S_angry_o", "S_angry_c"],
}


def _lin(p, name, x):
    return Fn.linear(x, p[name + ".weight"], p[name + ".bias"])


def factored_step(p, x, h, c, mode):
    """stylenet/model.py:115-155: returns (h', c')."""
    pre = []
    for g, s in zip("ifoc", MODE_S[mode]):
        v = _lin(p, "V_" + g, x)
        v = _lin(p, s, v)
        pre.append(_lin(p, "U_" + g, v) + _lin(p, "W_" + g, h))
    i_t, f_t, o_t = torch.sigmoid(pre[0]), torch.sigmoid(pre[1]), torch.sigmoid(pre[2])
    c_tilda = torch.tanh(pre[3])
    c = f_t * c + i_t * c_tilda
    h = o_t * c                     # no tanh on the cell (model.py:153)
    return h, c


def lstmcell_step(p, x, h, c):
    """nn.LSTMCell (nic/model.py:52,77): gates i,f,g,o; h = o*tanh(c)."""
    gates = Fn.linear(x, p["lstm.weight_ih"], p["lstm.bias_ih"]) + \
        Fn.linear(h, p["lstm.weight_hh"], p["lstm.bias_hh"])
    i, f, g, o = gates.chunk(4, 1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def batch_sizes(lengths):
    return [sum(1 for l in lengths if l > t) for t in range(max(lengths))]


def _run(step, emb_w, out_w, out_b, captions, lengths, features, tf_mask, hidden_size,
         drop_mask=None):
    """Shared loop of stylenet/model.py:157-196 / nic/model.py:81-115.

    drop_mask: optional [B, T, E] multiplicative dropout mask (already scaled by 1/(1-p)).
    Returns packed logits [N, V] and the list of predicted token tensors per step.
    """
    B = captions.size(0)
    embeddings = emb_w[captions]                                  # B(captions)
    if drop_mask is not None:
        embeddings = embeddings * drop_mask                       # self.dropout(embeddings)
    if features is not None:
        embeddings = torch.cat((features.unsqueeze(1), embeddings), 1)
    bs = batch_sizes(lengths)
    assert len(tf_mask) == len(bs)
    h = torch.zeros(B, hidden_size, dtype=emb_w.dtype)
    c = torch.zeros(B, hidden_size, dtype=emb_w.dtype)
    hiddens, preds = [], []
    predicted = captions[:, 0:1]
    for i, b in enumerate(bs):
        if tf_mask[i]:
            x = embeddings[:b, i, :]
        else:
            x = emb_w[predicted][:b, 0, :]                        # no dropout on the feedback
        h, c = h[:b], c[:b]
        h, c = step(x, h, c)
        hiddens.append(h)
        output = Fn.linear(h, out_w, out_b)
        predicted = output.max(1)[1].unsqueeze(1)
        preds.append(predicted)
    hiddens = torch.cat(hiddens, 0)
    return Fn.linear(hiddens, out_w, out_b), preds


def factored_lstm_forward(p, captions, lengths, features, tf_mask, mode="factual", drop_mask=None):
    H = p["W_i.weight"].shape[0]
    return _run(lambda x, h, c: factored_step(p, x, h, c, mode), p["B.weight"], p["C.weight"],
                p["C.bias"], captions, lengths, features, tf_mask, H, drop_mask)[0]


def stacked_factored_lstm_forward(p, captions, lengths, features, tf_mask, mode="factual", num_layers=2):
    """capnet.stacked.StackedFactoredLSTM.forward restated on the CPU. PARITY UNPINNED: the reference ignores num_layers
    (stylenet/model.py:37); this is SURVEY App. A-1's definition -- layer l > 0 is the factored cell of model.py:115-155
    on the hidden state of the layer below at the same step (parameters `V1_i`, `S1_fi`, ...), the top layer feeds C,
    the loop is model.py:157-196 -- and what it pins is the GPU engine to an independent statement of that definition."""
    H = p["W_i.weight"].shape[0]
    B = captions.size(0)
    emb_w = p["B.weight"]
    embeddings = emb_w[captions]
    if features is not None:
        embeddings = torch.cat((features.unsqueeze(1), embeddings), 1)
    bs = batch_sizes(lengths)
    hs = [torch.zeros(B, H, dtype=emb_w.dtype) for _ in range(num_layers)]
    cs = [torch.zeros(B, H, dtype=emb_w.dtype) for _ in range(num_layers)]
    sfx = {"factual": "f", "happy": "happy_", "sad": "sad_", "angry": "angry_"}[mode]
    hiddens = []
    predicted = captions[:, 0:1]
    for i, b in enumerate(bs):
        x = embeddings[:b, i, :] if tf_mask[i] else emb_w[predicted][:b, 0, :]
        for l in range(num_layers):
            tag = "" if l == 0 else str(l)
            pre = []
            for g in "ifoc":
                v = _lin(p, "V%s_%s" % (tag, g), x)
                v = _lin(p, "S%s_%s%s" % (tag, sfx, g), v)
                pre.append(_lin(p, "U%s_%s" % (tag, g), v) + _lin(p, "W%s_%s" % (tag, g), hs[l][:b]))
            i_t, f_t, o_t = torch.sigmoid(pre[0]), torch.sigmoid(pre[1]), torch.sigmoid(pre[2])
            cs[l] = f_t * cs[l][:b] + i_t * torch.tanh(pre[3])
            hs[l] = o_t * cs[l]
            x = hs[l]
        hiddens.append(x)
        predicted = Fn.linear(x, p["C.weight"], p["C.bias"]).max(1)[1].unsqueeze(1)
    return Fn.linear(torch.cat(hiddens, 0), p["C.weight"], p["C.bias"])


def lstm_forward(p, captions, lengths, features, tf_mask, drop_mask=None):
    H = p["lstm.weight_hh"].shape[1]
    return _run(lambda x, h, c: lstmcell_step(p, x, h, c), p["embed.weight"], p["linear.weight"],
                p["linear.bias"], captions, lengths, features, tf_mask, H, drop_mask)[0]


def packed_targets(captions, lengths):
    """pack_padded_sequence(captions, lengths, batch_first=True)[0]
    (stylenet/train_multitask.py:377-379)."""
    bs = batch_sizes(lengths)
    return torch.cat([captions[:b, t] for t, b in enumerate(bs)], 0)


# ---------------------------------------------------------------------------------------
# attention decoder: DecoderFactoredLSTMAtt.forward, stylenet/model_att.py:238-305
# ---------------------------------------------------------------------------------------
MODE_ATT = {"factual": "attention", "happy": "attention_happy", "sad": "attention_sad",
            "angry": "attention_angry"}


def attention_step(p, prefix, feat, h):
    """Attention.forward, stylenet/model_att.py:51-70."""
    att1 = _lin(p, prefix + ".encoder_att", feat)                 # recomputed every step there
    att2 = _lin(p, prefix + ".decoder_att", h)
    att = _lin(p, prefix + ".full_att", torch.relu(att1 + att2.unsqueeze(1))).squeeze(2)
    alpha = torch.softmax(att, dim=1)
    return (feat * alpha.unsqueeze(2)).sum(dim=1), alpha


def _att_run(p, step, emb_name, out_name, att, captions, lengths, features, tf_mask, drop_mask):
    """Shared loop of stylenet/model_att.py:238-305 and nic/model_att.py:152-202."""
    B = captions.size(0)
    feat = features.reshape(B, -1, features.size(-1))
    P = feat.size(1)
    emb_w = p[emb_name + ".weight"]
    embeddings = emb_w[captions]
    if drop_mask is not None:
        embeddings = embeddings * drop_mask
    bs = batch_sizes(lengths)
    mean = feat.mean(dim=1)
    h = _lin(p, "init_h", mean)
    c = _lin(p, "init_c", mean)
    hiddens, alpha_list = [], []
    predicted = captions[:, 0:1]
    for i, b in enumerate(bs):
        awe, alpha = attention_step(p, att, feat[:b], h[:b])
        gate = torch.sigmoid(_lin(p, "f_beta", h[:b]))
        awe = gate * awe
        if tf_mask[i]:
            x = embeddings[:b, i, :]
        else:
            x = emb_w[predicted][:b, 0, :]
        h, c = step(torch.cat([x, awe], dim=1), h[:b], c[:b])
        hiddens.append(h)
        alpha_list.append((b, i, alpha))
        predicted = _lin(p, out_name, h).max(1)[1].unsqueeze(1)
    # alphas[:b, i, :] = alpha without in-place writes (keeps autograd simple)
    cols = []
    for b, i, alpha in alpha_list:
        cols.append(torch.cat([alpha, torch.zeros(B - b, P, dtype=alpha.dtype)], 0).unsqueeze(1))
    alphas = torch.cat(cols, 1)
    return _lin(p, out_name, torch.cat(hiddens, 0)), alphas


def factored_att_forward(p, captions, lengths, features, tf_mask, mode="factual", drop_mask=None):
    """DecoderFactoredLSTMAtt.forward. Returns (packed logits [N, V], alphas [B, max(lengths), P]).
    `captions` / `lengths` are what the training loop passes: captions[:, :-1] and lengths - 1
    (train_multitask_att.py:402-408)."""
    return _att_run(p, lambda x, h, c: factored_step(p, x, h, c, mode), "B", "C", MODE_ATT[mode],
                    captions, lengths, features, tf_mask, drop_mask)


def lstm_att_forward(p, captions, lengths, features, tf_mask, drop_mask=None):
    """nic DecoderRNNAtt.forward (nic/model_att.py:152-202): the same loop around nn.LSTMCell."""
    return _att_run(p, lambda x, h, c: lstmcell_step(p, x, h, c), "embed", "linear", "attention",
                    captions, lengths, features, tf_mask, drop_mask)


def att_loss(logits, alphas, targets, alpha_c=1.0):
    """criterion(outputs, targets) + alpha_c * ((1 - alphas.sum(dim=1))**2).mean()
    (stylenet/train_multitask_att.py:409-411)."""
    return Fn.cross_entropy(logits, targets) + alpha_c * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
