"""Command line driver with the reference CLI's sub-commands, flags, validation and output naming
(cli/src/args.rs:11-216, cli/src/main.rs:46-239), on top of libkmeans_hip.

    python -m kmeans_gpu_amd.cli reduce  -i img.png -c 8 [-a kmeans|octree] [-m replace|dither|meld] [-o out.png]
    python -m kmeans_gpu_amd.cli find    -i img.png -p "#050505,#ffffff,#ff0000"|palette.png [-m ...] [-o out.png]
    python -m kmeans_gpu_amd.cli palette -i img.png -c 8 [-a ...] [-s 40] [-o out.png]

Image decoding/encoding (the `image` crate in the reference) is done with Pillow.  One flag the reference does not have:
`--devices 0,1,...` (before the sub-command) runs the same operation over several GPUs of the node (kmg_group_*: the image
tiled in row bands, same bytes).
"""
import argparse
import os
import re
import sys
import time

import numpy as np

from . import Algorithm, Group, ImageProcessor, ReduceMode

_PALETTE_RE = re.compile(r"^#[0-9a-fA-F]{6}(?:,#[0-9a-fA-F]{6})*$")     # args.rs:184
_MODES = {"replace": ReduceMode.Replace, "dither": ReduceMode.Dither, "meld": ReduceMode.Meld}
_ALGOS = {"kmeans": Algorithm.Kmeans, "octree": Algorithm.Octree}


def validate_k(s):                                       # args.rs:160-171
    try:
        k = int(s)
    except ValueError:
        k = 0
    if k < 1:
        raise argparse.ArgumentTypeError("k must be an integer higher than 0.")
    return k


def validate_filename(s):                                # args.rs:173-179
    if len(s) > 4 and (s.endswith(".png") or s.endswith(".jpg")):
        return s
    raise argparse.ArgumentTypeError("Only support png or jpg files.")


def parse_colors(s):                                     # args.rs:218-231
    return np.array([[int(c[1:3], 16), int(c[3:5], 16), int(c[5:7], 16), 255] for c in s.split(",")], np.uint8)


def parse_palette(path):                                 # args.rs:197-216
    from PIL import Image
    px = np.array(Image.open(path).convert("RGBA")).reshape(-1, 4)
    if px.shape[0] > 512:
        raise argparse.ArgumentTypeError("Trying to load a palette with more than 512 colors")
    colors = sorted(set(map(tuple, px)))
    if len(colors) < px.shape[0]:
        raise argparse.ArgumentTypeError("Trying to load a palette with recuring colors")
    return np.array(colors, np.uint8)


def validate_palette(s):                                 # args.rs:181-195
    if _PALETTE_RE.match(s):
        return parse_colors(s)
    if len(s) > 4 and (s.endswith(".png") or s.endswith(".jpg")) and os.path.exists(s):
        return parse_palette(s)
    raise argparse.ArgumentTypeError('The palette should be a path to an image file, or defined as "#RRGGBB,#RRGGBB,#RRGGBB"')


def _load(path):
    from PIL import Image
    return np.array(Image.open(path).convert("RGBA"))      # image::open(..).to_rgba8()


def _save(path, rgba):
    from PIL import Image
    Image.fromarray(rgba, "RGBA").save(path)


def reduce_file_path(k, algo, mode, output, inp):        # main.rs:127-153
    if output:
        return output
    stem = os.path.splitext(os.path.basename(inp))[0]
    return os.path.join(os.path.dirname(inp), f"{stem}-reduce-c{k}-{algo}-{mode}.png")


def palette_file_path(k, inp, output, algo, size):       # main.rs:155-182
    if output:
        return output
    stem = os.path.splitext(os.path.basename(inp))[0]
    return os.path.join(os.path.dirname(inp), f"{stem}-palette-c{k}-{algo}-s{size}.png")


def find_file_path(mode, output, inp):                   # main.rs:184-219
    if output:
        return output
    stem, ext = os.path.splitext(os.path.basename(inp))
    millis = int(time.time() * 1000)
    return os.path.join(os.path.dirname(inp), f"{stem}-find-{mode}-{millis}{ext}")


def validate_size(s):                                       # args.rs:36-38 value_parser!(u32).range(1..=60)
    try:
        v = int(s)
    except ValueError:
        raise argparse.ArgumentTypeError(f"invalid value '{s}': not an integer")
    if not 1 <= v <= 60:
        raise argparse.ArgumentTypeError(f"{v} is not in 1..=60")
    return v


def validate_devices(s):
    try:
        devices = [int(v) for v in s.split(",")]
    except ValueError:
        devices = []
    if not devices or min(devices) < 0:
        raise argparse.ArgumentTypeError("--devices takes a comma separated list of device ordinals, e.g. 0,1")
    return devices


def main(argv=None):
    ap = argparse.ArgumentParser(prog="kmeans-hip", description="k-means colour quantisation on MI355X")
    ap.add_argument("--devices", type=validate_devices, default=None,
                    help="HIP device ordinals, e.g. 0,1,2,3: tile the image over several GPUs (no counterpart in the reference)")
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("palette", help="Create an image with the dominant colors of the input")
    p.add_argument("-c", "--colorcount", type=validate_k, required=True)
    p.add_argument("-i", "--input", type=validate_filename, required=True)
    p.add_argument("-o", "--output", type=validate_filename)
    p.add_argument("-a", "--algo", choices=list(_ALGOS), default="kmeans")
    p.add_argument("-s", "--size", type=validate_size, default=40)
    f = sub.add_parser("find", help="Replace the colors of the input with the closest ones of a palette")
    f.add_argument("-i", "--input", type=validate_filename, required=True)
    f.add_argument("-o", "--output", type=validate_filename)
    f.add_argument("-p", "--palette", type=validate_palette, required=True)
    f.add_argument("-m", "--mode", choices=list(_MODES), default="replace")
    r = sub.add_parser("reduce", help="Reduce the number of colors of the input")
    r.add_argument("-c", "--colorcount", type=validate_k, required=True)
    r.add_argument("-i", "--input", type=validate_filename, required=True)
    r.add_argument("-o", "--output", type=validate_filename)
    r.add_argument("-a", "--algo", choices=list(_ALGOS), default="kmeans")
    r.add_argument("-m", "--mode", choices=list(_MODES), default="replace")
    args = ap.parse_args(argv)

    image = _load(args.input)
    with (Group(devices=args.devices) if args.devices else ImageProcessor()) as proc:
        if args.command == "palette":                    # main.rs:46-72
            colors = proc.palette(args.colorcount, image, _ALGOS[args.algo])
            out = np.repeat(np.repeat(colors[None, :, :], args.size, axis=0), args.size, axis=1)   # main.rs:221-239
            _save(palette_file_path(args.colorcount, args.input, args.output, args.algo, args.size), out)
            print("Palette: " + ",".join(f"#{c[0]:02X}{c[1]:02X}{c[2]:02X}" for c in colors))
        elif args.command == "find":                     # main.rs:74-98
            out = proc.find(image, args.palette, _MODES[args.mode])
            _save(find_file_path(args.mode, args.output, args.input), out)
        else:                                            # main.rs:100-125
            out = proc.reduce(args.colorcount, image, _ALGOS[args.algo], _MODES[args.mode])
            _save(reduce_file_path(args.colorcount, args.algo, args.mode, args.output, args.input), out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
