// TEST INFRASTRUCTURE ONLY.  main() for the reference's OWN benchmark sources (sdrbench/mainbench.cpp, parserbench.cpp, compiled
// where they lie, unchanged) built against qt_adapter/shadow/: its Decimators / DecimatorsIF / FI / FF members are the GPU
// classes.  Stands in for appbench/main.cpp, minus the qtwebapp file logger (MainBench only stores that pointer).
#include <QCoreApplication>
#include <QObject>
#include <QTimer>
#include "mainbench.h"

int main(int argc, char* argv[])
{
    QCoreApplication a(argc, argv);
    QCoreApplication::setApplicationName("sdrangelbench on libsdrx");
    ParserBench parser;
    parser.parse(a);
    MainBench m(nullptr, parser, &a);
    QObject::connect(&m, SIGNAL(finished()), &a, SLOT(quit()));
    QTimer::singleShot(0, &m, SLOT(run()));
    return a.exec();
}
