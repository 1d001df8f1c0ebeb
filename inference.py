#!/usr/bin/env python3
"""AuroraCap inference on MI355X - same argument surface and stdout contract as the reference CLI
(rese1f/aurora inference.py:29-98): prints one caption string.

    python inference.py --model_path <xtuner-format dir> --visual_input clip.mp4 --num_frm 8 \
        --token_kept_ratio 0.3 --max_new_tokens 256

Tokenizer and PyAV frame decoding use the same third-party packages the reference uses; resize / crop / normalise
(bit-identical to the reference's CLIPImageProcessor call) and the three hot stages run in libaurora_hip.so.
`--synthetic` replaces checkpoint, tokenizer and video by the seeded synthetic clip of SURVEY 8d (prints ids).
"""
import argparse
import os.path as osp
import sys

import numpy as np
import torch


from aurora_amd.preprocess import read_video_pyav  # noqa: E402  (frame sampling contract of load_video.py:31-71)


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    # the reference CLI's argument surface (inference.py:31-40): same flags, types and defaults
    for flag, typ, default, text in (
            ("model_path", str, "wchai/AuroraCap-7B-IMG-xtuner", "checkpoint directory (xtuner layout) or hub id"),
            ("prompt", str, "Describe the video in detail.", "instruction given to the model"),
            ("visual_input", str, "output.png", "video (.mp4) or image (.png / .jpg) file"),
            ("num_frm", int, 8, "frames sampled uniformly from a video"),
            ("token_kept_ratio", float, 0.8, "fraction of visual tokens kept by the per-layer token merge"),
            ("temperature", float, 0.0, "sampling temperature (greedy only on this path)"),
            ("top_p", float, 1.0, "nucleus mass (unused: greedy)"),
            ("num_beams", int, 1, "beam count (only 1 is implemented)"),
            ("max_new_tokens", int, 2048, "generation budget")):
        parser.add_argument("--" + flag, type=typ, default=default, help=text)
    parser.add_argument('--host_preprocess', action='store_true', help='resize/normalise on the host with CLIPImageProcessor (PIL) instead of the HIP input stage')
    parser.add_argument('--synthetic', action='store_true', help='seeded synthetic weights / clip / prompt ids (no checkpoint needed)')
    return parser


def main():
    args = build_parser().parse_args()
    if args.num_beams != 1:
        sys.exit("error: only greedy decoding is implemented on the MI355X path (--num_beams 1)")

    from aurora_amd.model import AuroraModel, build_prompt, process_text

    if args.synthetic:
        from aurora_amd import synthetic as S
        from aurora_amd.engine import AuroraCapEngine
        cfg = S.AURORACAP_7B
        w = {"vit": S.vit_weights(cfg["vit"]), "projector": S.projector_weights(1280, 4096), "llm": S.llm_weights(cfg["llm"])}
        eng = AuroraCapEngine(cfg, w, max_frames=args.num_frm, max_batch=1, max_ctx=30 + args.num_frm * 729 + args.max_new_tokens + 64,
                              max_new_tokens=args.max_new_tokens)
        model = AuroraModel(eng, eos_token_id=None)
        data = {"pixel_values": S.frames(args.num_frm, 0).unsqueeze(0), "input_ids": torch.tensor([S.prompt_ids(args.num_frm, 0)])}
        tokenizer = None
    else:
        from transformers import AutoTokenizer, CLIPImageProcessor
        if not osp.isdir(args.model_path):
            from huggingface_hub import snapshot_download
            args.model_path = snapshot_download(repo_id=args.model_path)
        # input size / patch size come from the checkpoint's own config; the reference writes the AuroraCap-7B values
        # (size=378, crop_size=378, inference.py:58-63) into its processor call, which is what the release's config says too
        from aurora_amd.checkpoint import vit_config
        vcfg = vit_config(osp.join(args.model_path, "visual_encoder"))
        image_size, patch = vcfg["image_size"], vcfg["patch_size"]
        if args.host_preprocess:      # the reference's host path (PIL); default is the bit-identical HIP input stage
            try:
                image_processor = CLIPImageProcessor.from_pretrained("laion/CLIP-ViT-bigG-14-laion2B-39B-b160k", size=image_size, crop_size=image_size)
            except OSError:           # no hub access: that repository's preprocessor_config.json holds the class defaults (OpenAI CLIP mean / std, bicubic)
                image_processor = CLIPImageProcessor(size=image_size, crop_size=image_size)

            def to_pixels(frames):
                return image_processor(list(frames), return_tensors='pt')['pixel_values'].to(dtype=torch.float16)
        else:
            from aurora_amd.preprocess import FramePreprocessor
            gpu_pre = FramePreprocessor(image=image_size)

            def to_pixels(frames):
                return gpu_pre(torch.from_numpy(np.ascontiguousarray(np.stack(list(frames)))).cuda())
        tokenizer = AutoTokenizer.from_pretrained(args.model_path, trust_remote_code=True, padding_side='right')
        data = dict()
        if args.visual_input.endswith('mp4'):
            video_frames = read_video_pyav(args.visual_input, args.num_frm)
            data["pixel_values"] = to_pixels(video_frames).unsqueeze(0)
            n_img = len(video_frames)
        elif args.visual_input.endswith('png') or args.visual_input.endswith('jpg'):
            from PIL import Image
            image = np.asarray(Image.open(args.visual_input).convert("RGB"))       # the processor's do_convert_rgb
            data["pixel_values"] = to_pixels([image])
            n_img = 1
        else:
            sys.exit("error: --visual_input must end with mp4, png or jpg")
        data["input_ids"] = process_text(build_prompt(args.prompt, n_img), tokenizer)
        # The engine's capacity comes from THIS request: read_video_pyav may return num_frm + 1 frames and the
        # prompt length is only known after tokenising, so size the context from the spliced length, not from the flags.
        n_text = int((data["input_ids"] != -200).sum())
        per_frame = (image_size // patch) ** 2                                      # upper bound: token_kept_ratio = 1.0 keeps all patches
        max_ctx = -(-(n_text + n_img * per_frame + args.max_new_tokens) // 64) * 64
        model = AuroraModel.from_pretrained(args.model_path, max_frames=max(n_img, 1), max_ctx=max_ctx,
                                            max_new_tokens=args.max_new_tokens)

    model.visual_encoder.reset_tome_r(args.token_kept_ratio)
    output = model(data, mode="inference")
    cont = model.llm.generate(**output, do_sample=False, temperature=args.temperature, top_p=args.top_p,
                              num_beams=args.num_beams, max_new_tokens=args.max_new_tokens)
    if tokenizer is None:
        print(cont[0].tolist())
    else:
        print(tokenizer.batch_decode(cont, skip_special_tokens=True)[0])


if __name__ == "__main__":
    main()
