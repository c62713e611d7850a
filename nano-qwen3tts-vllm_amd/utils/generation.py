"""Talker prompt construction: token ids, speaker, language and optional voice-clone material become the
input embeddings of the talker prefill plus the per-frame text hiddens that are added while decoding
(reference: P/utils/generation.py:18-379 `prepare_inputs`, :382-409 `generate_speaker_prompt`, :412-515
`generate_icl_prompt`).

Layout of one request (D = talker hidden, every row is text-side + codec-side):

    [instruct]                      proj(text_emb(instruct ids))                             optional
    role                            proj(text_emb(ids[:3]))                                  "<|im_start|>assistant\\n"
    codec prefix, P-1 rows          [pad .. pad, bos](text)  +  codec_emb(prefix[:-1])
        prefix = think/nothink, think_bos, [language], think_eos, [speaker], codec_pad, codec_bos      (P ids)
    then one of
      streaming      proj(ids[3:4]) + codec_emb(codec_bos)                  trailing = proj(ids[4:-5]) ++ eos
      non-streaming  (proj(ids[3:-5]) ++ eos) + codec_emb(codec_pad) rows,
                     pad + codec_emb(codec_bos)                             trailing = pad
      in-context     text/codec rows of the reference utterance (generate_icl_prompt)

The batch is left-padded with zero rows (mask 0 there); trailing hiddens are right-padded with the pad
embedding.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import torch


def _speaker_row(config, speaker, voice_clone_prompt, voice_clone_spk_embeds, index, input_embedding, device, dtype):
    if voice_clone_spk_embeds is not None:
        use = voice_clone_prompt["x_vector_only_mode"][index] or voice_clone_prompt["icl_mode"][index]
        return voice_clone_spk_embeds[index] if use else None
    if speaker is None or speaker == "":
        return None
    table = config.talker_config.spk_id
    if speaker.lower() not in table:
        raise NotImplementedError(f"Speaker {speaker} not implemented")
    return input_embedding(torch.tensor(table[speaker.lower()], device=device, dtype=dtype))


def _language_id(config, language, speaker):
    tc = config.talker_config
    assert language is not None
    lang = language.lower()
    if lang == "auto":
        lid = None
    else:
        if lang not in tc.codec_language_id:
            raise NotImplementedError(f"Language {language} not implemented")
        lid = tc.codec_language_id[lang]
    # a dialect speaker overrides the language token for Chinese / unspecified text
    if lang in ("chinese", "auto") and speaker not in ("", None):
        dialect = tc.spk_is_dialect.get(speaker.lower(), False)
        if dialect is not False:
            lid = tc.codec_language_id[dialect]
    return lid


def prepare_inputs(config, input_ids: list, instruct_ids: Optional[list] = None, ref_ids: Optional[list] = None,
                   voice_clone_prompt: Optional[dict] = None, languages: Optional[list] = None,
                   speakers: Optional[list] = None, non_streaming_mode: bool = False, *, text_embedding: Callable,
                   input_embedding: Callable, text_projection: Callable, device: Union[torch.device, str],
                   voice_clone_spk_embeds: Optional[list] = None, generate_speaker_prompt_fn: Optional[Callable] = None,
                   generate_icl_prompt_fn: Optional[Callable] = None):
    """Returns (talker_input_embeds [B,Lmax,D], trailing_text_hiddens [B,Tmax,D], tts_pad_embed [1,1,D],
    talker_attention_mask [B,Lmax])."""
    device = torch.device(device) if isinstance(device, str) else device
    tc = config.talker_config
    B = len(input_ids)
    if voice_clone_prompt is not None and voice_clone_spk_embeds is None:
        if generate_speaker_prompt_fn is None:
            raise ValueError("voice_clone_prompt is set but neither voice_clone_spk_embeds nor generate_speaker_prompt_fn "
                             "was provided")
        voice_clone_spk_embeds = generate_speaker_prompt_fn(voice_clone_prompt)
    speakers = [None] * B if speakers is None else speakers
    languages = ["auto"] * B if languages is None else languages

    def text(ids):                                   # token ids -> talker-width text rows
        return text_projection(text_embedding(ids))

    def codec(ids, dtype):
        return input_embedding(torch.tensor(ids, device=device, dtype=dtype))

    prompts, trailing, pad = [], [], None
    for i, (ids, language, speaker) in enumerate(zip(input_ids, languages, speakers)):
        dt = ids.dtype
        parts = []
        if instruct_ids is not None and instruct_ids[i] is not None:
            ins = instruct_ids[i]
            parts.append(text((ins.unsqueeze(0) if ins.dim() == 1 else ins).to(device)))
        spk = _speaker_row(config, speaker, voice_clone_prompt, voice_clone_spk_embeds, i, input_embedding, device, dt)
        lid = _language_id(config, language, speaker)
        bos, eos, pad = text(torch.tensor([[config.tts_bos_token_id, config.tts_eos_token_id, config.tts_pad_token_id]],
                                          device=device, dtype=dt)).chunk(3, dim=1)
        head = [tc.codec_nothink_id, tc.codec_think_bos_id, tc.codec_think_eos_id] if lid is None else \
               [tc.codec_think_id, tc.codec_think_bos_id, lid, tc.codec_think_eos_id]
        prefix = [codec([head], dt)] + ([spk.view(1, 1, -1)] if spk is not None else []) + \
                 [codec([[tc.codec_pad_id, tc.codec_bos_id]], dt)]
        prefix = torch.cat(prefix, dim=1)                                   # [1, P, D]
        P = prefix.shape[1]
        parts.append(text(ids[:, :3]))
        parts.append(torch.cat((pad.expand(-1, P - 2, -1), bos), dim=1) + prefix[:, :-1])

        icl = (voice_clone_prompt is not None and voice_clone_prompt.get("ref_code") is not None
               and voice_clone_prompt["icl_mode"][i])
        if icl:
            if generate_icl_prompt_fn is None:
                raise ValueError("ICL mode requested but generate_icl_prompt_fn was not provided")
            if ref_ids is None or ref_ids[i] is None:
                raise ValueError(f"ICL mode requires ref_ids, but ref_ids[{i}] is None. Please provide ref_text when "
                                 "creating voice_clone_prompt or when calling generate_voice_clone.")
            rows, tail = generate_icl_prompt_fn(text_id=ids[:, 3:-5], ref_id=ref_ids[i][:, 3:-2],
                                                ref_code=voice_clone_prompt["ref_code"][i].to(device),
                                                tts_pad_embed=pad, tts_eos_embed=eos,
                                                non_streaming_mode=non_streaming_mode)
            parts.append(rows)
        elif non_streaming_mode:
            body = ids[:, 3:-5]
            n = body.shape[1] + 1
            parts.append(torch.cat((text(body), eos), dim=1) + codec([[tc.codec_pad_id] * n], dt))
            parts.append(pad + codec([[tc.codec_bos_id]], dt))
            tail = pad
        else:
            parts.append(text(ids[:, 3:4]) + prefix[:, -1:])
            tail = torch.cat((text(ids[:, 4:-5]), eos), dim=1)
        prompts.append(torch.cat(parts, dim=1).squeeze(0))
        trailing.append(tail.squeeze(0))

    lens = [p.shape[0] for p in prompts]
    Lmax, D = max(lens), prompts[0].shape[-1]
    embeds = prompts[0].new_zeros(B, Lmax, D)
    mask = torch.zeros(B, Lmax, dtype=torch.long, device=embeds.device)
    for i, p in enumerate(prompts):                                         # left padding
        embeds[i, Lmax - lens[i]:] = p
        mask[i, Lmax - lens[i]:] = 1
    Tmax = max(t.shape[0] for t in trailing)
    tails = pad.reshape(1, 1, D).expand(B, Tmax, D).clone()
    for i, t in enumerate(trailing):
        tails[i, :t.shape[0]] = t
    return embeds, tails, pad, mask


def generate_speaker_prompt(voice_clone_prompt: dict, device: Union[torch.device, str],
                            dtype: Optional[torch.dtype] = None) -> List[torch.Tensor]:
    """The reference speaker embeddings of a voice-clone prompt, moved to `device` (and `dtype`)."""
    device = torch.device(device) if isinstance(device, str) else device
    out = []
    for e in voice_clone_prompt["ref_spk_embedding"]:
        e = e.to(device)
        out.append(e if dtype is None else e.to(dtype))
    return out


def generate_icl_prompt(text_id: torch.Tensor, ref_id: torch.Tensor, ref_code: torch.Tensor, tts_pad_embed: torch.Tensor,
                        tts_eos_embed: torch.Tensor, non_streaming_mode: bool, config, text_embedding: Callable,
                        input_embedding: Callable, text_projection: Callable, code_predictor_embeddings: List[Callable],
                        device: Union[torch.device, str]):
    """In-context rows of a voice clone: the reference transcript and the new text on the text side, codec_bos
    and the summed 16 codebook embeddings of `ref_code` (columns 0..15 of the rows it holds) on the codec side.
    Returns (rows [1,T,D], trailing hiddens)."""
    device = torch.device(device) if isinstance(device, str) else device
    tc = config.talker_config
    words = torch.cat([text_projection(text_embedding(torch.cat([ref_id, text_id], dim=-1))), tts_eos_embed], dim=1)
    books = []
    for g in range(tc.num_code_groups):
        table = input_embedding if (g == 0 or g - 1 >= len(code_predictor_embeddings)) else code_predictor_embeddings[g - 1]
        books.append(table(ref_code[:, g:g + 1]))
    frame = torch.cat(books, dim=1).sum(1).unsqueeze(0)                     # [1, rows of ref_code, D]
    bos = input_embedding(torch.tensor([[tc.codec_bos_id]], device=device, dtype=text_id.dtype))
    codes = torch.cat([bos, frame], dim=1)
    T, C = words.shape[1], codes.shape[1]
    if non_streaming_mode:
        pads = input_embedding(torch.tensor([[tc.codec_pad_id] * T], device=device, dtype=text_id.dtype))
        return torch.cat([words + pads, codes + tts_pad_embed], dim=1), tts_pad_embed
    if T > C:
        return words[:, :C] + codes, words[:, C:]
    words = torch.cat([words] + [tts_pad_embed] * (C - T), dim=1)
    return words + codes, tts_pad_embed
