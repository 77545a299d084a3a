"""Text-prompt twin of the pipeline: ``EEG2Video/pipelines/pipeline_tuneavideo.py:315-412`` (caller
``train_finetune_videodiffusion.py:331-335``: ``validation_pipeline(prompt, generator=..., latents=..., **validation_data)``).

Same class name, ``__call__`` kwargs (``prompt``, ``negative_prompt``, ``num_videos_per_prompt``), checks and error types.  The CLIP
text encoder is outside the accelerated path, so the conditioning enters in one of two ways:

* ``prompt`` is a ``[B,77,768]`` tensor of precomputed prompt embeddings (``negative_prompt`` then a ``[1 or B,77,768]`` tensor, or
  ``pipe.negative_embeddings``) -- nothing but this library runs;
* ``prompt`` is a ``str`` / ``list`` as in the reference, and the pipeline was given a ``tokenizer`` and a ``text_encoder`` (host-side
  torch modules with the ``transformers`` CLIP interface): ``_encode_prompt`` (:149-243) tokenises and encodes exactly as the
  reference does, the empty prompt giving the unconditional embedding.

From there on it is the EEG pipeline's loop (``pipeline.py``): the fused device loop for deterministic DDIM, else stepped.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import torch

from .pipeline import TuneAVideoPipeline as _EEGPipeline, TuneAVideoPipelineOutput  # noqa: F401


class TuneAVideoPipeline(_EEGPipeline):
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):                           # :46-60 (argument order of the reference)
        super().__init__(vae=vae, tokenizer=tokenizer, unet=unet, scheduler=scheduler)
        self.text_encoder = text_encoder

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, text_encoder=None, **kwargs):
        base = _EEGPipeline.from_pretrained(pretrained_model_path, **kwargs)
        return cls(vae=base.vae, text_encoder=text_encoder, tokenizer=base.tokenizer, unet=base.unet, scheduler=base.scheduler)

    def _encode_text(self, prompts: List[str], device, max_length=None):
        """tokenizer + text encoder on the host side, as :152-179 / :211-229."""
        if self.tokenizer is None or self.text_encoder is None:
            raise ValueError("a `str` / `list` prompt needs the pipeline's `tokenizer` and `text_encoder` (host-side CLIP modules, outside "
                             "the accelerated path); pass precomputed embeddings as a [B,77,768] tensor instead")
        tok = self.tokenizer(prompts, padding="max_length", max_length=max_length or self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt")
        cfg = getattr(self.text_encoder, "config", None)
        mask = tok.attention_mask.to(device) if getattr(cfg, "use_attention_mask", False) else None
        return self.text_encoder(tok.input_ids.to(device), attention_mask=mask)[0].float()

    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):   # :149-243
        if isinstance(prompt, torch.Tensor):
            emb = prompt.to(device).float()
            if emb.dim() == 2:
                emb = emb[None]
            batch_size = emb.shape[0]
        else:
            batch_size = len(prompt) if isinstance(prompt, list) else 1
            emb = self._encode_text(prompt if isinstance(prompt, list) else [prompt], device)
        bs, seq_len, _ = emb.shape
        emb = emb.repeat(1, num_videos_per_prompt, 1).view(bs * num_videos_per_prompt, seq_len, -1)              # :182-184
        if not do_classifier_free_guidance:
            return emb, None
        if isinstance(prompt, torch.Tensor):
            neg = negative_prompt if negative_prompt is not None else self.negative_embeddings
            if not isinstance(neg, torch.Tensor):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")                                                                  # :191-195
            neg = neg.to(device).float().reshape(-1, seq_len, emb.shape[-1])
            if neg.shape[0] not in (1, batch_size):
                raise ValueError(f"`negative_prompt`: has batch size {neg.shape[0]}, but `prompt`: has batch size {batch_size}. Please "
                                 "make sure that passed `negative_prompt` matches the batch size of `prompt`.")      # :198-203
            if neg.shape[0] == 1:
                neg = neg.expand(batch_size, -1, -1)
        else:
            if negative_prompt is None:
                uncond_tokens = [""] * batch_size                                                                     # :189-190
            elif type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond_tokens = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                                 " the batch size of `prompt`.")
            else:
                uncond_tokens = negative_prompt
            neg = self._encode_text(uncond_tokens, device, max_length=seq_len)
        neg = neg.repeat(1, num_videos_per_prompt, 1).view(batch_size * num_videos_per_prompt, seq_len, -1)          # :232-234
        return emb, neg

    def check_inputs(self, prompt, height, width, callback_steps):                                                   # :270-285
        if not isinstance(prompt, (str, list, torch.Tensor)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        super().check_inputs(torch.empty(0), height, width, callback_steps)

    @torch.no_grad()
    def __call__(
        self,
        prompt: Union[str, List[str], torch.Tensor],
        video_length: Optional[int],
        height: Optional[int] = None,
        width: Optional[int] = None,
        num_inference_steps: int = 50,
        guidance_scale: float = 7.5,
        negative_prompt: Optional[Union[str, List[str], torch.Tensor]] = None,
        num_videos_per_prompt: Optional[int] = 1,
        eta: float = 0.0,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "tensor",
        return_dict: bool = True,
        callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
        callback_steps: Optional[int] = 1,
        **kwargs,
    ):
        height = height or self.unet.config.sample_size * self.vae_scale_factor                                      # :336-337
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)                                                     # :340
        emb, neg = self._encode_prompt(prompt, self._execution_device, num_videos_per_prompt, guidance_scale > 1.0, negative_prompt)
        # the embeddings are already repeated per prompt: the EEG pipeline's loop takes them as they are (model = None)
        return _EEGPipeline.__call__(self, None, emb, video_length, height=height, width=width, num_inference_steps=num_inference_steps,
                                     guidance_scale=guidance_scale, negative_prompt=neg, num_videos_per_eeg=1, eta=eta,
                                     generator=generator, latents=latents, output_type=output_type, return_dict=return_dict,
                                     callback=callback, callback_steps=callback_steps, **kwargs)
