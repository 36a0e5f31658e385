"""Generate tests/golden/framing.json by importing the reference's OWN framing helper
(/root/reference/src/utils/stream_helper.py:19-99) in the build container: fixed strings in, the file bytes the
reference writes out. Run once here (the reference does not exist on the GPU box); the JSON is data only
(hex strings of inputs and outputs + get_downsampled_shape samples)."""
import importlib.util
import json
import os
import random
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_stream_helper", "/root/reference/src/utils/stream_helper.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rnd = random.Random(1234)


def blob(n):
    return bytes(rnd.getrandbits(8) for _ in range(n))


def main():
    out = {"i_frames": [], "p_frames": [], "shapes": []}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bin")
        for h, w, ny, nz in [(540, 960, 0, 0), (1080, 1920, 1, 0), (64, 64, 0, 5), (1152, 1920, 1336, 88),
                             (2160, 3840, 7001, 409), (1, 1, 3, 3), (65535, 65536, 257, 255)]:
            y, z = blob(ny), blob(nz)
            ref.encode_i(h, w, y, z, path)
            data = open(path, "rb").read()
            assert ref.decode_i(path) == (h, w, y, z) and ref.filesize(path) == len(data)
            out["i_frames"].append({"height": h, "width": w, "y": y.hex(), "z": z.hex(), "file": data.hex()})
        for n in (0, 1, 4, 1023, 65537):
            s = blob(n)
            ref.encode_p(s, path)
            data = open(path, "rb").read()
            assert ref.decode_p(path) == s
            out["p_frames"].append({"string": s.hex(), "file": data.hex()})
    for h, w, p, r in [(1080, 1920, 64, 1), (540, 960, 64, 1), (1080, 1920, 16, 1), (1152, 1920, 64, 1), (720, 1280, 64, 1),
                       (2160, 3840, 64, 1), (480, 832, 64, 1), (1, 1, 64, 1), (65, 129, 16, 2), (1080, 1920, 64, 2),
                       (100, 200, 3, 1), (7, 7, 2, 1)]:
        out["shapes"].append({"args": [h, w, p, r], "out": list(ref.get_downsampled_shape(h, w, p, r))})
    with open(os.path.join(HERE, "framing.json"), "w") as f:
        json.dump(out, f)
    print("framing.json:", len(out["i_frames"]), "I files,", len(out["p_frames"]), "P files,", len(out["shapes"]), "shapes")


if __name__ == "__main__":
    main()
