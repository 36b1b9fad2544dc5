"""Result export: ``saveResultClass`` of
PyREMOT/library/saveResult.py:12-47 - a list to ``saveFile.txt`` (one ``str(element)`` per line)
or a list of rows plus a header to ``saveFile.csv``, both in the current directory, with the
reference's behaviour for non-list input (a printed message, no file).  The optional ``path``
argument is an addition; its default is the reference's fixed file name."""
import csv


class saveResultClass:
    def __init__(self):
        pass

    def saveListToText(data, path="saveFile.txt"):
        if not isinstance(data, list):
            print("data is not a list")
        else:
            with open(path, "w") as textfile:
                for element in data:
                    textfile.write(str(element) + "\n")

    def saveListToCSV(data, headerList, path="saveFile.csv"):
        if not isinstance(data, list):
            print("data is not a list")
        else:
            with open(path, "w", newline="") as f:
                write = csv.writer(f)
                write.writerow(headerList)
                write.writerows(data)
