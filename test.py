"""Entry point with the reference's name and command line (`python test.py --i_frame_model_path ... --model_path ...
--test_config ... --cuda 1 --worker N --write_stream 0|1 --output_path ...`, /root/reference/test.py:36-81,665-791).
A wrapper only: the evaluation harness lives in lssvc_amd/harness.py."""
from lssvc_amd.harness import main

if __name__ == "__main__":
    main()
