"""Parameter files in the reference's grammar (`Comment : key = value`, `#` comments;
reference src/Common/Parameters.cpp:75-152) -> dict of strings."""


def read_params_file(path):
    out = {}
    with open(path, "r", errors="replace") as f:
        for line in f:
            line = line.split("#", 1)[0]
            if "=" not in line:
                continue
            if ":" in line:
                line = line.split(":", 1)[1]
            key, val = line.split("=", 1)
            out[key.strip()] = val.strip()
    return out
