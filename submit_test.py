"""Counterpart of the reference's submit_test.py (:5-28): builds the 8-GPU evaluation command line and runs it.
Paths come from the environment instead of being hard-coded to one cluster:
    LSSVC_I_MODELS / LSSVC_P_MODELS  space-separated checkpoint lists (q1..q4), LSSVC_TEST_CONFIG (recommend_test_config.json),
    LSSVC_OUTPUT (/output/LSSVC_IP32), LSSVC_STREAM_PATH (/output/out_bin), LSSVC_WORKERS (8), LSSVC_DEVICES (0,...,7)."""
import os
import subprocess
import sys


def build_command(env=os.environ):
    experiment_name = "LSSVC_IP32"
    i_models = env.get("LSSVC_I_MODELS", "").split()
    p_models = env.get("LSSVC_P_MODELS", "").split()
    if not i_models or len(i_models) != len(p_models):
        raise SystemExit("set LSSVC_I_MODELS and LSSVC_P_MODELS to equally long, space-separated checkpoint lists")
    workers = env.get("LSSVC_WORKERS", "8")
    return [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "test.py"),
            "--i_frame_model_name", "IntraSS", "--i_frame_model_path", *i_models, "--model_path", *p_models,
            "--test_config", env.get("LSSVC_TEST_CONFIG", "recommend_test_config.json"),
            "--cuda", "1", "--worker", workers,
            "--cuda_device", env.get("LSSVC_DEVICES", ",".join(str(i) for i in range(int(workers)))),
            "--write_stream", "0", "--output_path", env.get("LSSVC_OUTPUT", "/output/" + experiment_name),
            "--stream_path", env.get("LSSVC_STREAM_PATH", "/output/out_bin"), "--save_decoded_mv", "0",
            "--model_name", "LSSVC_extend"]


if __name__ == "__main__":
    cmd = build_command()
    print(" ".join(cmd))
    sys.exit(subprocess.call(cmd))
