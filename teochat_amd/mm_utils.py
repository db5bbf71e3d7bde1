"""Host-side multimodal utilities (mirror of the reference's videollava/mm_utils.py surface).

tokenizer_image_token  -- image-token packing of the tokenized prompt (mm_utils.py:43-62), bit-exact
KeywordsStoppingCriteria -- stop test on the generated tail (mm_utils.py:73-104)
process_images / expand2square / get_model_name_from_path -- mm_utils.py:14-40,65-71
All integer / string work; it stays on the host (the tokenizer is third-party and host-side).
"""
import torch

from .constants import IMAGE_TOKEN_INDEX


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenize the text between "<image>" markers separately and join the pieces with one sentinel each.
    A BOS emitted by the tokenizer is kept once at the very front and stripped from every later piece."""
    pieces = [list(tokenizer(part).input_ids) for part in prompt.split("<image>")]
    bos = getattr(tokenizer, "bos_token_id", None)
    lead = 1 if (pieces and pieces[0] and pieces[0][0] == bos) else 0
    packed = pieces[0][:lead]
    for n, piece in enumerate(pieces):
        if n:
            packed.append(image_token_index)
        packed.extend(piece[lead:])
    if return_tensors is None:
        return packed
    if return_tensors == "pt":
        return torch.tensor(packed, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]


def expand2square(pil_img, background_color):
    """Pad a PIL image to a square canvas, centred (mm_utils.py:14-25)."""
    from PIL import Image
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_images(images, image_processor, model_cfg):
    if getattr(model_cfg, "image_aspect_ratio", None) != "pad":
        return image_processor(images, return_tensors="pt")["pixel_values"]
    fill = tuple(int(c * 255) for c in image_processor.image_mean)
    if getattr(image_processor, "engine", None) is not None:
        # the padding happens inside the device kernel (teo_preprocess_frames_pad): the square canvas is never built
        out = [image_processor.preprocess(im, return_tensors="pt", pad_rgb=fill)["pixel_values"][0] for im in images]
    else:
        out = [image_processor.preprocess(expand2square(im, fill), return_tensors="pt")["pixel_values"][0] for im in images]
    if all(o.shape == out[0].shape for o in out):
        return torch.stack(out, dim=0)
    return out


class KeywordsStoppingCriteria:
    """Callable with the transformers.StoppingCriteria calling convention: (output_ids, scores) -> bool.

    Stops when the tail of a row equals one of the keywords' token ids, or when the decoded tail
    (at most max_keyword_len new tokens) contains a keyword; a batch stops when every row does.
    `keyword_id_lists` lets the generator run the id-suffix test on the device (teo_decode_state.d_stop_ids).
    """

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]
        self.keyword_id_lists = []
        for kw in self.keywords:
            ids = list(tokenizer(kw).input_ids)
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.keyword_id_lists.append(ids)
        self.max_keyword_len = max((len(i) for i in self.keyword_id_lists), default=0)
        self.keyword_ids = [torch.tensor(i) for i in self.keyword_id_lists]

    def call_for_batch(self, output_ids, scores=None, **kwargs):
        row = output_ids[0].tolist()
        for ids in self.keyword_id_lists:
            if row[-len(ids):] == ids:
                return True
        window = min(len(row) - self.start_len, self.max_keyword_len)
        tail = row[-window:]           # window == 0 selects the whole row, exactly like the reference's [-0:] slice
        text = self.tokenizer.batch_decode([tail], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores=None, **kwargs):
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))
