"""Sentinels and token strings of the path.

The names and values are an interface (prompts, checkpoints and callers of the reference depend on them): they equal
videollava/constants.py:7-10,17,24.  Only the entries the image path uses carry meaning here; the video / patch / start-end
tokens exist so that `from videollava.constants import ...` keeps working for code written against the reference.
"""

# label value the loss ignores, and the input-id sentinel that marks where one image's visual tokens are spliced in
IGNORE_INDEX, IMAGE_TOKEN_INDEX = -100, -200

# prompt-side markers: replace_video_token() turns every DEFAULT_VIDEO_TOKEN into DEFAULT_IMAGE_TOKEN x T
DEFAULT_IMAGE_TOKEN, DEFAULT_VIDEO_TOKEN = "<image>", "<video>"
IMAGE_PLACEHOLDER, VIDEO_PLACEHOLDER = "<image-placeholder>", "<video-placeholder>"

# optional wrapping tokens (mm_use_im_start_end / mm_use_im_patch_token configurations; unused by TEOChat's checkpoints)
DEFAULT_IMAGE_PATCH_TOKEN = DEFAULT_VIDEO_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<im_start>", "<im_end>"
DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN = "<vid_start>", "<vid_end>"

# dataset-side limits of the reference's training collator (kept for import compatibility)
MAX_IMAGE_LENGTH, MAX_VIDEO_LENGTH, PAD_LENGTH = 16, 1, 620
CONTROLLER_HEART_BEAT_EXPIRATION = 30
WORKER_HEART_BEAT_INTERVAL = 15

LOGDIR = "."
