"""Config reader with the reference's behaviour (utils.py:3-11): any failure prints the
exception and 'Parsing config file error', then exits with status 3."""
import json


def read_config(file_name):
    try:
        with open(file_name, "r") as f:
            return json.load(f)
    except Exception as e:  # noqa: BLE001 - the reference catches everything
        print(e)
        print("Parsing config file error")
        raise SystemExit(3)
