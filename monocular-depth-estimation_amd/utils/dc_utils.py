"""Minimal video I/O under the reference's module path ``utils.dc_utils`` so its drivers import cleanly
(reference: utils/dc_utils.py:19-86).  Decoding/encoding is outside the accelerated path; these helpers only wrap
whichever of OpenCV / imageio is installed and raise a clear error otherwise."""
import numpy as np


def read_video_frames(video_path, process_length=-1, target_fps=-1, max_res=-1):
    """uint8 RGB frames ``[N, H, W, 3]`` and the effective fps: frames are strided to ``target_fps`` and scaled down so the
    longer side is at most ``max_res``."""
    try:
        import cv2
    except ImportError as e:
        raise ImportError("read_video_frames needs OpenCV (cv2) to decode video files") from e
    cap = cv2.VideoCapture(video_path)
    src_fps = cap.get(cv2.CAP_PROP_FPS)
    h, w = int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)), int(cap.get(cv2.CAP_PROP_FRAME_WIDTH))
    if max_res > 0 and max(h, w) > max_res:
        s = max_res / max(h, w)
        h, w = round(h * s), round(w * s)
    fps = src_fps if target_fps < 0 else target_fps
    stride = max(round(src_fps / fps), 1)
    frames, i = [], 0
    while True:
        ok, frame = cap.read()
        if not ok or (0 < process_length <= len(frames)):
            break
        if i % stride == 0:
            frame = cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)
            if frame.shape[:2] != (h, w):
                frame = cv2.resize(frame, (w, h))
            frames.append(frame)
        i += 1
    cap.release()
    return np.stack(frames, axis=0), fps


def save_video(frames, output_video_path, fps=10, is_depths=False, grayscale=False):
    """Write frames, or depth maps normalised to 8 bits and coloured with the inferno table, with imageio's ffmpeg writer
    (reference: utils/dc_utils.py:72-86)."""
    try:
        import imageio
    except ImportError as e:
        raise ImportError("save_video needs imageio with the ffmpeg plugin") from e
    writer = imageio.get_writer(output_video_path, fps=fps, macro_block_size=1, codec="libx264", ffmpeg_params=["-crf", "18"])
    if is_depths:       # normalisation and colour table on the device (video_depth_anything/visualize.py)
        from video_depth_anything.visualize import colorize_depth
        for g in colorize_depth(frames, grayscale=grayscale):
            writer.append_data(g)
    else:
        for f in frames:
            writer.append_data(f)
    writer.close()
