"""
Extract the experiment data file of BASELINE config 1 from the reference's shipped archive
(docs/diffusion_processes/data.zip, member data/linear/15/0.npz: the OU process of
docs/diffusion_processes/README.md:26-43 -- decay 0.5, q = 1, dt = 0.01, T = 1001, 32 training and 8 test observations,
sigma = 0.1) into tests/golden/linear_15_0.npz, byte for byte.  Data only; read through `exp_io.load_exp_data`
(the mirror of exp_dp_utils.py:108-125) by tests/test_oracle_models.py and tests/test_gpu_api.py.

Run ONLY in the build container (needs /root/reference):  python tests/golden/extract_exp_data.py
"""
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
ARCHIVE = "/root/reference/docs/diffusion_processes/data.zip"
MEMBERS = {"data/linear/15/0.npz": "linear_15_0.npz"}


def main():
    with zipfile.ZipFile(ARCHIVE) as z:
        for member, name in MEMBERS.items():
            with open(os.path.join(HERE, name), "wb") as fh:
                fh.write(z.read(member))
            print("wrote", name)


if __name__ == "__main__":
    main()
